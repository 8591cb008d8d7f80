# one GPU call: the whole GPU suite, then the side workloads of bench.py (logs under gpurun_out/)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2_pytest_gpu.log
timeout -k 10 300 python bench.py --workload tiny-saint --steps 3 --no-cpu-baseline > gpurun_out/r2_bench_tiny_saint.json 2> gpurun_out/r2_bench_tiny_saint.err; echo "tiny-saint rc=$?"
timeout -k 10 600 python bench.py --workload cfg4-saint --steps 10 --warmup 2 > gpurun_out/r2_bench_cfg4_saint.json 2> gpurun_out/r2_bench_cfg4_saint.err; echo "cfg4-saint rc=$?"; cat gpurun_out/r2_bench_cfg4_saint.json | cut -c1-1500
timeout -k 10 600 python bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_cfg5.json 2> gpurun_out/r2_bench_cfg5.err; echo "cfg5 rc=$?"; cat gpurun_out/r2_bench_cfg5.json | cut -c1-1200
