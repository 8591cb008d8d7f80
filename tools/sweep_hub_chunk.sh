#!/bin/bash
# Developer tool: cfg5 (RMAT, bf16 storage) and cfg4 steps for several long-segment chunk sizes.
mkdir -p gpurun_out/chunk
for c in ${CHUNKS:-64 128 256 512}; do
  AMPCONV_HUB_CHUNK=$c python3 bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-gemm --no-extra \
    > gpurun_out/chunk/cfg5_$c.json 2> gpurun_out/chunk/cfg5_$c.err || { echo "chunk $c failed"; continue; }
  python3 - <<PY
import json
d = json.loads(open('gpurun_out/chunk/cfg5_$c.json').read().strip().splitlines()[-1])
print('chunk $c', d['value'], d['ms_per_step'], d.get('kernels_ms'), d['roofline']['frac'], flush=True)
PY
done
