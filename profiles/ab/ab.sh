#!/bin/bash
# A/B of two builds of the library on ONE box: alternates them under the same command.  usage: bash profiles/ab/ab.sh <rounds> <command...>
rounds=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
L=image-preprocessing-pipeline_amd/libmi_ipp.so
cp $L /tmp/libA.so
for r in $(seq 1 $rounds); do
  cp /tmp/libA.so $L; echo "A: $("$@" 2>/dev/null | tail -n 1)"
  cp profiles/ab/libmi_ipp_B.so $L; echo "B: $("$@" 2>/dev/null | tail -n 1)"
done
cp /tmp/libA.so $L
