#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: kernel-trace stats of the headline run, the two TCC traffic passes
# (separate: FETCH_SIZE / WRITE_SIZE do not fit one pass), kernel stats of cfg5 and the SQ counter passes.
# Counters are collected with --kernel-trace only (no other trace domain).
#   usage: tools/run_profiles.sh <tag> [part]     part: all (default) | bench | counters
tag=${1:-r05}
part=${2:-all}
d=gpurun_out/prof_$tag
mkdir -p $d
export TMPDIR=/tmp
if [ "$part" != counters ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-gemm --no-extra > $d/bench_stats.log 2> $d/bench_stats.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $d -o fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-gemm --no-extra > $d/bench_fetch.log 2> $d/bench_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $d -o write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-gemm --no-extra > $d/bench_write.log 2> $d/bench_write.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o cfg5 -- python3 bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $d/bench_cfg5.log 2> $d/bench_cfg5.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o saint -- python3 bench.py --workload cfg4-saint --steps 10 --warmup 2 --no-cpu-baseline > $d/bench_saint.log 2> $d/bench_saint.err || exit 1
rm -f $d/saint_kernel_trace.csv
fi
if [ "$part" != bench ]; then
{
echo "## plane-format edge kernels (cfg4 shape at 0.4 of its size: uniform graph 400 k / 4 M, L=20 D=256 H=8; statistics hand-off on)"
bash tools/prof_sq.sh ${tag}_planes 400000 4000000 20 256 8 --planes
echo "## fp32 edge kernels (uniform graph 100 k / 1 M, L=20 D=256 H=8)"
bash tools/prof_sq.sh ${tag}_f32
bash tools/prof_fp32_pipe.sh ${tag}_pipe_f32
echo "## bf16 edge kernels (R-MAT, cfg5 shape at 1/4 size)"
bash tools/prof_sq.sh ${tag}_bf16 524288 10000000 20 256 8 --bf16 --rmat
bash tools/prof_fp32_pipe.sh ${tag}_pipe_bf16 524288 10000000 20 256 8 --bf16 --rmat
echo "## workgroup-per-unit kernels (AMPGCN default shape L=40 D=100 H=2), bound-carrying entry points: two fp16 planes"
bash tools/prof_sq.sh ${tag}_block_scaled 100000 1000000 40 100 2 --scaled
echo "## the same shape through the fp32 entry points: three bf16 planes"
bash tools/prof_sq.sh ${tag}_block 100000 1000000 40 100 2
echo "## fp32 projections, scaled two-plane mode (8M rows, D=256)"
bash tools/prof_sq_proj.sh ${tag}_proj_scaled 8000000 256 --iters=2 --no-lib --scaled
echo "## fp32 projections, six-product form"
bash tools/prof_sq_proj.sh ${tag}_proj_6p 8000000 256 --iters=2 --no-lib
echo "## bf16 projections"
bash tools/prof_sq_proj.sh ${tag}_proj_bf16 8000000 256 --iters=2 --no-lib --bf16
} > $d/sq_summary.txt 2>&1
# the raw counter CSVs are large (gpurun copies back at most 64 MiB): the summary is what is kept
rm -rf gpurun_out/sq_r4/${tag}_* gpurun_out/sq_r5/${tag}_*
fi
ls $d | head -40
tail -60 $d/sq_summary.txt
