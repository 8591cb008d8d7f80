#!/bin/bash
# Developer tool: hub-graph tests, then the bf16 / fp32 edge kernels on the cfg5 graph and one cfg5 bench line.
set -e
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "hub or bf16_storage or config5 or stable_sort" 2>&1 | tail -3
echo "== rmat bf16"; python3 tools/bench_kernels.py 2097152 40000000 20 256 8 --bf16 --rmat 2>&1 | tail -3
echo "== uniform bf16"; python3 tools/bench_kernels.py --bf16 2>&1 | tail -3
echo "== rmat fp32 1M/20M"; python3 tools/bench_kernels.py 1048576 20000000 20 256 8 --rmat 2>&1 | tail -3
python3 bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-gemm --no-extra > gpurun_out/cfg5_scramble.json 2> gpurun_out/cfg5_scramble.err
cat gpurun_out/cfg5_scramble.json
