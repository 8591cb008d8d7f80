#!/bin/bash
# Developer tool: a few end-of-round measurements in one GPU call.
mkdir -p gpurun_out/misc
echo "== class-default shape on a Cora-sized graph (block family)"
python3 tools/bench_kernels.py 2708 10556 40 100 2 2>&1 | grep -E "fwd_edge|bwd_edge"
echo "== cfg5 with rocBLAS (default) / hipBLASLt bf16 GEMMs"
for pref in 0 1; do
  TORCH_BLAS_PREFER_HIPBLASLT=$pref python3 bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-gemm --no-extra \
    > gpurun_out/misc/cfg5_blaslt$pref.json 2> gpurun_out/misc/cfg5_blaslt$pref.err
  python3 - <<PY
import json
d = json.loads(open('gpurun_out/misc/cfg5_blaslt$pref.json').read().strip().splitlines()[-1])
print('hipblaslt=$pref', d['value'], d['ms_per_step'], d['node_phase_ms'], d['node_phase_mfma'])
PY
done
echo "== forced-distributed line with the saint record"
AMPCONV_BENCH_FORCE_DIST=1 python3 bench.py --no-cpu-baseline --no-alt-gemm > gpurun_out/misc/dist1.json 2> gpurun_out/misc/dist1.err
python3 - <<PY
import json
d = json.loads(open('gpurun_out/misc/dist1.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('saint'), d.get('allreduce_ms'))
PY
