#!/bin/bash
# SQ counter passes over tools/bench_kernels.py for one build / switch combination (developer tool).
#   tools/prof_sq.sh <tag> [bench_kernels args]   (environment: AMPCONV_LIB_PATH, AMPCONV_SRC_NT4, ...)
# Counters only (no trace domains besides the kernel trace), two passes of <= 8 SQ counters.
tag=$1; shift
out=gpurun_out/${SQ_DIR:-sq_r5}/$tag
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $out -o p1 -- python3 tools/bench_kernels.py "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $out -o p2 -- python3 tools/bench_kernels.py "$@" > $out/p2.log 2>&1
python3 tools/summarize_sq2.py $out
