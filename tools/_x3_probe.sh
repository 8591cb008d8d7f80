set -e
A="100000 1000000 40 100 2"
for v in x3nc x3nl; do
AMPCONV_LIB_PATH=build/variants/libampconv_$v.so python tools/bench_kernels.py $A > gpurun_out/x3_$v.log 2>&1
echo $v; tail -n 3 gpurun_out/x3_$v.log
done
python tools/bench_kernels.py $A > gpurun_out/x3_base.log 2>&1; tail -n 3 gpurun_out/x3_base.log
