#!/bin/bash
# developer tool: correctness + speed of every proj_rows tile shape (AMPCONV_PROJ_ROWS), then the stamped build
for v in ${VARIANTS:-0 1 2}; do
  echo "== AMPCONV_PROJ_ROWS=$v"
  AMPCONV_PROJ_ROWS=$v timeout -k 10 300 python -m pytest tests/test_gpu_proj.py -x -q -k "rows" 2>&1 | tail -2
  AMPCONV_PROJ_ROWS=$v timeout -k 10 200 python tools/bench_proj.py --only=rows --no-lib 2>&1 | grep native
  if [ -f build/variants/libampconv_PSTAMP.so ]; then AMPCONV_PROJ_ROWS=$v timeout -k 10 120 python tools/stamp_proj.py 2>&1 | grep -v amdgpu.ids; fi
done
