#!/bin/bash
# Two rocprofv3 SQ counter passes over one NCC batch on the C5 grid, per kernel (profiles/pmc_sq.py).
# usage: bash profiles/sq_ncc.sh <out.txt>
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MI_IPP_PROBES=1 MI_NCC_SERIAL_MIPS=1
timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/sqA -o pmc -- python3 profiles/ncc_batch_probe.py 1 > gpurun_out/sqA.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/sqB -o pmc -- python3 profiles/ncc_batch_probe.py 1 > gpurun_out/sqB.log 2>&1
python3 profiles/pmc_sq.py gpurun_out/sqA gpurun_out/sqB > "$out"
rm -rf gpurun_out/sqA gpurun_out/sqB
