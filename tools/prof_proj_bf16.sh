#!/bin/bash
# Counter passes over ONE bf16 projection product of tools/bench_proj.py (developer tool).
#   tools/prof_proj_bf16.sh <tag> <qkv|out|dx|dwin|dwo> [rows D]
# Counters only (kernel trace + --pmc), SQ in two passes, the two TCC byte counters in passes of their own.
tag=$1; which=$2; shift 2
out=gpurun_out/sq_r4/$tag
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -o $name -- python3 tools/bench_proj.py ${ARGS} --bf16 --no-lib --only=$which --iters=3 > $out/$name.log 2>&1; }
ARGS="$*"
run p1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
run p2 SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
run p3 FETCH_SIZE
run p4 WRITE_SIZE
python3 tools/summarize_sq2.py $out
