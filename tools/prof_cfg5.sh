d=gpurun_out/prof_r03
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o cfg5 -- python3 bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $d/bench_cfg5.log 2> $d/bench_cfg5.err || exit 1
TAGS="rmat" bash tools/prof_rmat_passes.sh 2>&1 | grep -v "^[EW]2026"
