#!/bin/bash
# FP32-pipe occupancy of the fp32 edge kernels from counters (developer tool).
#   tools/prof_fp32_pipe.sh <tag> [bench_kernels args]
# One counter pass (kernel trace + --pmc only) over tools/bench_kernels.py; the summary divides the matrix-pipe busy
# cycles and the VALU-active cycles by SIMDs x wall cycles.
tag=$1; shift
out=gpurun_out/sq_r4/$tag
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $out -o p1 -- python3 tools/bench_kernels.py "$@" > $out/p1.log 2>&1
python3 tools/summarize_pipe.py $out
