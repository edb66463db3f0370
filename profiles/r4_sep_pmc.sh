#!/bin/bash
# HBM traffic of the single-pass separable kernel (C2-sized volume, 9 x 9 x 15 taps): FETCH_SIZE and WRITE_SIZE in separate passes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_SF -o pmc -- python3 profiles/separable_time.py > /dev/null 2>&1
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_SW -o pmc -- python3 profiles/separable_time.py > /dev/null 2>&1
python3 profiles/pmc_summary.py gpurun_out/pmc_SF gpurun_out/pmc_SW gpurun_out/r04_sep_pmc_traffic.json | grep -i "sep3d\|conv3d"
rm -rf gpurun_out/pmc_SF gpurun_out/pmc_SW
