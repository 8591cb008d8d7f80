#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: kernel-trace stats of the headline run, the two TCC traffic passes
# (separate: FETCH_SIZE / WRITE_SIZE do not fit one pass), kernel stats of cfg5 and the SQ counter passes.
# Counters are collected with --kernel-trace only (no other trace domain).   usage: tools/run_profiles.sh <tag>
tag=${1:-r03}
d=gpurun_out/prof_$tag
mkdir -p $d
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-gemm --no-extra > $d/bench_stats.log 2> $d/bench_stats.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $d -o fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-gemm --no-extra > $d/bench_fetch.log 2> $d/bench_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $d -o write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-gemm --no-extra > $d/bench_write.log 2> $d/bench_write.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o cfg5 -- python3 bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $d/bench_cfg5.log 2> $d/bench_cfg5.err || exit 1
bash tools/prof_sq.sh ${tag}_final > $d/sq_summary.txt 2>&1
bash tools/prof_sq_proj.sh ${tag}_proj 8000000 256 --iters=2 --no-lib >> $d/sq_summary.txt 2>&1
ls $d | head -40
cat $d/sq_summary.txt
