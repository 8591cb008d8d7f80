#!/bin/bash
# Developer tool: does the pitch of the Q|K|V rows matter on raw R-MAT node ids?  (bf16, cfg5 graph)
for pad in 0 64 32 8; do
  echo "== rowpad $pad"; python3 tools/bench_kernels.py 2097152 40000000 20 256 8 --bf16 --rmat --rowpad=$pad 2>&1 | grep -E "fwd_edge|bwd_edge"
done
echo "== rowpad 64 uniform"; python3 tools/bench_kernels.py --bf16 --rowpad=64 2>&1 | grep -E "fwd_edge|bwd_edge"
