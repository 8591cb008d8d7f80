set -e
timeout -k 10 500 python -m pytest tests/test_gpu_block_scaled.py -x -q -m gpu > gpurun_out/x3_t4.log 2>&1 || (tail -n 40 gpurun_out/x3_t4.log; exit 1)
tail -n 2 gpurun_out/x3_t4.log
W="--workload ampgcn-default --steps 5 --no-cpu-baseline --no-alt-gemm --no-extra"
for v in default xhA xhC; do
  if [ $v = default ]; then python bench.py $W > gpurun_out/x3_b_$v.log 2>&1; else AMPCONV_LIB_PATH=build/variants/libampconv_$v.so python bench.py $W > gpurun_out/x3_b_$v.log 2>&1; fi
  echo $v; grep -o '"kernels_ms": {[^}]*}' gpurun_out/x3_b_$v.log; grep -o '"ms_per_step": [0-9.]*' gpurun_out/x3_b_$v.log | head -1
done
