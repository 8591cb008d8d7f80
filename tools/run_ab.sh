set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r2_pytest_parity.log 2>&1; echo "pytest rc=$?" 
tail -3 gpurun_out/r2_pytest_parity.log
for v in default src2w bperm; do
  if [ $v = default ]; then unset AMPCONV_LIB_PATH; else export AMPCONV_LIB_PATH=$PWD/build/variants/libampconv_$v.so; fi
  echo "=== $v" >> gpurun_out/r2_ab.log
  timeout -k 10 300 python tools/bench_kernels.py >> gpurun_out/r2_ab.log 2>&1
done
unset AMPCONV_LIB_PATH
echo "=== default, AMPCONV_SRC_NT4=0" >> gpurun_out/r2_ab.log
AMPCONV_SRC_NT4=0 timeout -k 10 300 python tools/bench_kernels.py >> gpurun_out/r2_ab.log 2>&1
grep -E "===|fwd_edge|bwd_edge" gpurun_out/r2_ab.log
