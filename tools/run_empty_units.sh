#!/bin/bash
# Developer tool: what does a unit with no edge cost?  2 M nodes, 1 000 edges: every launch is 16.8 M empty unit waves.
for flags in "--bf16" ""; do
  echo "== empty units $flags"; python3 tools/bench_kernels.py 2097152 1000 20 256 8 $flags --no-stats 2>&1 | tail -3
done
echo "== 1M nodes empty (fp32)"; python3 tools/bench_kernels.py 1048576 1000 20 256 8 --no-stats 2>&1 | tail -3
