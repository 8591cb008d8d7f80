"""profiles/<tag>_sq_counters.md from `rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
--kernel-trace --output-format csv -d <dir> -o <mode> -- python3 tools/bench_kernels.py ...` runs.

    python tools/summarize_sq.py gpurun_out/sq r01"""
import collections
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = (('fwd', r'\bfwd_(mfma|bf16)'), ('bwd_dst', r'\bbwd_dst_(mfma|bf16)'), ('bwd_src', r'\bbwd_src_(mfma|bf16)'))


def main():
    d, tag = sys.argv[1], sys.argv[2]
    rows = []
    for mode in ('exact', 'bf16'):
        acc = {k: collections.defaultdict(list) for k, _ in KERNELS}
        dur = {k: [] for k, _ in KERNELS}
        seen = set()
        for r in csv.DictReader(open(os.path.join(d, f'{mode}_counter_collection.csv'))):
            for k, pat in KERNELS:
                if re.search(pat, r['Kernel_Name']):
                    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
                    key = (k, r['Dispatch_Id'])
                    if key not in seen:
                        seen.add(key)
                        dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        for k, _ in KERNELS:
            c = {n: sum(v) / len(v) for n, v in acc[k].items()}
            t = sum(dur[k]) / len(dur[k]) * 1e-9
            clock = c['GRBM_GUI_ACTIVE'] / 8 / t            # 8 XCDs count in parallel
            wc = c['SQ_WAVE_CYCLES']
            rows.append((mode, k, t * 1e3, clock / 1e9, c['SQ_WAIT_ANY'] / wc, c['SQ_WAIT_INST_ANY'] / wc,
                         c['SQ_ACTIVE_INST_ANY'] / wc, c['SQ_ACTIVE_INST_VALU'] / wc, c['SQ_ACTIVE_INST_LDS'] / wc,
                         c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (t * clock)))
    with open(os.path.join(ROOT, 'profiles', f'{tag}_sq_counters.md'), 'w') as f:
        f.write(f'# SQ counters of the edge kernels (rocprofv3 --pmc, {tag} final kernels)\n\n')
        f.write('Command: `rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY '
                'SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace '
                '--output-format csv -- python3 tools/bench_kernels.py 100000 1000000 20 256 8 --mode=exact` (and '
                '`--bf16`).  100 k nodes / 1 M edges, L=20, D=256, H=8; profiled runs are a few % slower than '
                'un-profiled ones.  Summarised by `tools/summarize_sq.py`.\n\n')
        f.write('Fractions are of SQ_WAVE_CYCLES (per-wave quad-cycles); `MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES / 1024 '
                'SIMDs / (duration x clock); clock = GRBM_GUI_ACTIVE / 8 / duration.\n\n')
        f.write('| mode | kernel | ms | clock GHz | wait (s_waitcnt) | issue stall | issuing | of which VALU | LDS | '
                'MFMA pipe busy |\n|---|---|---|---|---|---|---|---|---|---|\n')
        for r in rows:
            f.write('| %s | %s | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f |\n' % r)
        f.write('\nKernels: exact = `fwd_mfma_t4`, `bwd_dst_mfma_t4`, `bwd_src_mfma_t4` (batched tails, statistics '
                'hand-off); bf16 = `fwd_bf16`, `bwd_dst_bf16`, `bwd_src_bf16`.\n')


if __name__ == '__main__':
    main()
