set -e
bash tools/prof_x3.sh xh_scaled --scaled > gpurun_out/x3_prof_scaled.log 2>&1
bash tools/prof_x3.sh x3_bf16 > gpurun_out/x3_prof_bf16.log 2>&1
grep -v "^#" gpurun_out/sq_r5/xh_scaled/summary.txt | tail -9
grep -v "^#" gpurun_out/sq_r5/x3_bf16/summary.txt | tail -9
