"""Per-kernel averages of the counter passes written by tools/prof_sq.sh (developer tool).

    python tools/summarize_sq2.py gpurun_out/sq_r2/<tag>
Fractions are of SQ_WAVE_CYCLES; `mfma_busy` = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (duration x clock),
clock = GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS)."""
import collections
import csv
import glob
import os
import re
import sys

KERNELS = (('fwd', r'\bfwd_(mfma|bf16|block|f16x2|x3|xh)'), ('bwd_dst', r'\bbwd_dst_(mfma|bf16|block|f16x2|x3|xh)'),
           ('bwd_src', r'\bbwd_src_(mfma|bf16|block|f16x2|x3|xh)'),
           ('proj_rows', r'proj_rows_(bf16_)?kernel'), ('proj_wgrad', r'proj_wgrad_(bf16_)?kernel'))


def load(path):
    acc = {k: collections.defaultdict(list) for k, _ in KERNELS}
    dur = {k: {} for k, _ in KERNELS}
    for r in csv.DictReader(open(path)):
        for k, pat in KERNELS:
            if re.search(pat, r['Kernel_Name']):
                acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
                dur[k][r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    out = {}
    for k, _ in KERNELS:
        if not dur[k]:
            continue
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        c['_ms'] = sum(dur[k].values()) / len(dur[k]) * 1e-6
        out[k] = c
    return out


def main():
    d = sys.argv[1]
    res = {}
    for f in sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)):
        for k, c in load(f).items():
            res.setdefault(k, {}).update({n: v for n, v in c.items() if n not in res.get(k, {}) or n == '_ms'})
    print(f'# {d}')
    for k, c in res.items():
        t = c['_ms'] * 1e-3
        clock = c['GRBM_GUI_ACTIVE'] / 8 / t
        wc = c['SQ_WAVE_CYCLES']
        f = lambda n: c.get(n, float('nan')) / wc
        for n in ('FETCH_SIZE', 'WRITE_SIZE'):          # KiB per launch (MI355X_MICROARCH.md: wide reads count half on gfx950)
            if n in c:
                print(f"{k:8s} {n} {c[n] * 1024 / 1e9:.2f} GB per launch" + (' (x2 for 16-byte-per-lane reads)' if n == 'FETCH_SIZE' else ''))
        if 'SQ_WAVE_CYCLES' not in c:
            continue
        print(f"{k:8s} {c['_ms']:7.2f} ms clock {clock / 1e9:.2f} GHz | wait {f('SQ_WAIT_ANY'):.2f} issue-stall {f('SQ_WAIT_INST_ANY'):.2f} "
              f"(lds {f('SQ_WAIT_INST_LDS'):.2f}) issuing {f('SQ_ACTIVE_INST_ANY'):.2f} valu {f('SQ_ACTIVE_INST_VALU'):.2f} "
              f"lds {f('SQ_ACTIVE_INST_LDS'):.2f} vmem {f('SQ_ACTIVE_INST_VMEM'):.2f} | mfma_busy "
              f"{c.get('SQ_VALU_MFMA_BUSY_CYCLES', float('nan')) / 1024 / (t * clock):.2f} coexec "
              f"{c.get('SQ_VALU_MFMA_COEXEC_CYCLES', float('nan')) / 1024 / (t * clock):.2f} | lds conflict/active "
              f"{c.get('SQ_LDS_BANK_CONFLICT', float('nan')) / max(c.get('SQ_LDS_IDX_ACTIVE', float('nan')), 1):.2f} | insts/wave-cycle "
              f"valu {c.get('SQ_INSTS_VALU', float('nan')) / wc:.3f} mfma {c.get('SQ_INSTS_MFMA', float('nan')) / wc:.4f} "
              f"lds {c.get('SQ_INSTS_LDS', float('nan')) / wc:.4f} | totals valu {c.get('SQ_INSTS_VALU', 0):.3g} mfma {c.get('SQ_INSTS_MFMA', 0):.3g} lds {c.get('SQ_INSTS_LDS', 0):.3g}")


if __name__ == '__main__':
    main()
