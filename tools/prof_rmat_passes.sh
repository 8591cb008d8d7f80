#!/bin/bash
# Developer tool: per-launch durations (main pass / long-segment pass / combine) of the bf16 edge kernels on the cfg5 graph.
out=gpurun_out/rmat_passes; mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
for tag in ${TAGS:-rmat rmat_compact}; do
  extra=""; [ $tag = rmat_compact ] && extra="--compact"; [ $tag = rmat_sortdeg ] && extra="--sortdeg"; [ $tag = rmat_shuffle ] && extra="--shuffle"
  rocprofv3 --kernel-trace --output-format csv -d $out -o $tag -- python3 tools/bench_kernels.py 2097152 40000000 20 256 8 --bf16 --rmat $extra > $out/$tag.log 2>&1
  grep -E 'fwd_edge|bwd_edge|compacted' $out/$tag.log
  python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open('$out/${tag}_kernel_trace.csv')))
import re
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seq = []
for r in rows:
    m = re.search(r'(fwd_bf16|bwd_dst_bf16|bwd_src_bf16|hub_combine_kernel|fwd_mfma_t4|bwd_dst_mfma_t4|bwd_src_mfma_t4)', r['Kernel_Name'])
    if m: seq.append((m.group(1), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
agg = collections.OrderedDict()
# launches alternate main, long-segment per call: key by (name, parity of occurrence)
cnt = collections.Counter()
for n, ms in seq:
    short = n
    k = (short, cnt[short] % 2 if 'combine' not in short else 0); cnt[short] += 1
    agg.setdefault(k, []).append(ms)
for (n, par), v in agg.items():
    print(f'$tag {n:26s} {"main" if par == 0 else "long"}  n={len(v):3d}  avg {sum(v) / len(v):8.3f} ms')
PY
done
