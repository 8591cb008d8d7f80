#!/bin/bash
# SQ counter passes over tools/bench_proj.py (the node-phase projection kernels); developer tool.
#   tools/prof_sq_proj.sh <tag> [bench_proj args]
tag=$1; shift
out=gpurun_out/sq_r4/$tag
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $out -o p1 -- python3 tools/bench_proj.py "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $out -o p2 -- python3 tools/bench_proj.py "$@" > $out/p2.log 2>&1
python3 tools/summarize_sq2.py $out
