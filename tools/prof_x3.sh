#!/bin/bash
# SQ counters + HBM traffic of the workgroup-per-unit kernels on the AMPGCN class-default shape (developer tool).
#   tools/prof_x3.sh <tag> [--scaled]    (environment: AMPCONV_BLOCK_X3, AMPCONV_LIB_PATH)
tag=${1:-x3}; shift
ARGS="100000 1000000 40 100 2 $@"           # e.g. --scaled: the bound-carrying entry points (two fp16 planes)
SQ_DIR=${SQ_DIR:-sq_r5} bash tools/prof_sq.sh $tag $ARGS || exit 1
out=gpurun_out/${SQ_DIR:-sq_r5}/$tag
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out -o fetch -- python3 tools/bench_kernels.py $ARGS > $out/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out -o write -- python3 tools/bench_kernels.py $ARGS > $out/write.log 2>&1 || exit 1
python3 tools/summarize_sq2.py $out > $out/summary.txt
cat $out/summary.txt
