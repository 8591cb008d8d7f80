#!/bin/bash
# kernel launches per step of a bench.py workload (rocprofv3 --kernel-trace --stats); developer tool.
#   tools/prof_launches.sh <tag> <workload> [steps]
tag=$1; wl=$2; steps=${3:-50}
d=gpurun_out/prof_$tag
mkdir -p $d
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o k -- python3 bench.py --workload $wl --steps $steps --warmup 5 --no-cpu-baseline --no-alt-gemm --no-extra > $d/bench.log 2> $d/bench.err
grep '^{' $d/bench.log | cut -c1-260
python3 - "$d" "$steps" <<'PY'
import csv, glob, sys
d, steps = sys.argv[1], int(sys.argv[2]) + 5
f = glob.glob(d + '/**/k_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['Calls']) for r in rows)
print('kernel launches per step: %.1f   device time per step: %.1f us' % (tot / steps, sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e3))
for r in rows[:45]:
    print('%5.1f /step %8.1f us avg  %s' % (int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, r['Name'][:120]))
PY
