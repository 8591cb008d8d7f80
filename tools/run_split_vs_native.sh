#!/bin/bash
# Developer tool: the edge kernels in the native fp32 mode and in the split-operand modes, timings + SQ counters.
mkdir -p gpurun_out/split
for m in exact bf16x6 bf16x9; do
  echo "== $m"; python3 tools/bench_kernels.py --mode=$m 2>&1 | tail -6
done
bash tools/prof_sq.sh split_x6 --mode=bf16x6 > gpurun_out/split/sq_x6.txt 2>&1
bash tools/prof_sq.sh split_exact --mode=exact > gpurun_out/split/sq_exact.txt 2>&1
