"""FP32-pipe occupancy per kernel from the pass of tools/prof_fp32_pipe.sh (developer tool).

On gfx950 the fp32-input MFMA and the ordinary VALU instructions execute on ONE pipe per SIMD (tools/coexec_probe.hip:
a wave of MFMAs next to a wave of v_fma takes the SUM of the two).  Its occupancy over a launch is therefore
    (SQ_VALU_MFMA_BUSY_CYCLES + VALU execution cycles) / (1024 SIMDs x wall cycles),
wall cycles = duration x clock, clock = GRBM_GUI_ACTIVE / 8 / duration.  Two readings of "VALU execution cycles" are
printed: SQ_ACTIVE_INST_VALU x 4 (the counter is in quad-cycles; it counts the cycles a wave has a VALU-class
instruction in execution, MFMAs included while they issue) and the instruction count x the measured issue cost
(non-MFMA VALU x 4.5 cycles, coexec_probe: 1 MFMA + k v_fma = 48 + 5 k cycles)."""
import collections
import csv
import glob
import os
import re
import sys

KERNELS = (('fwd', r'\bfwd_(mfma|bf16|block)'), ('bwd_dst', r'\bbwd_dst_(mfma|bf16|block)'), ('bwd_src', r'\bbwd_src_(mfma|bf16|block)'))


def main():
    d = sys.argv[1]
    acc = {k: collections.defaultdict(list) for k, _ in KERNELS}
    dur = {k: {} for k, _ in KERNELS}
    for f in sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)):
        for r in csv.DictReader(open(f)):
            for k, pat in KERNELS:
                if re.search(pat, r['Kernel_Name']):
                    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
                    dur[k][r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    print(f'# {d}')
    print('| kernel | ms | clock GHz | matrix pipe busy | VALU active (counter) | sum | non-MFMA VALU x 4.5 cycles | sum | '
          'VALU / MFMA instructions | transcendental |')
    print('|---|---|---|---|---|---|---|---|---|---|')
    for k, _ in KERNELS:
        if not dur[k]:
            continue
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        t = sum(dur[k].values()) / len(dur[k]) * 1e-9
        clock = c['GRBM_GUI_ACTIVE'] / 8 / t
        simd_cycles = 1024 * t * clock
        mfma = c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles
        valu_ctr = 4 * c['SQ_ACTIVE_INST_VALU'] / simd_cycles
        n_valu = c['SQ_INSTS_VALU'] - c['SQ_INSTS_MFMA']
        valu_est = 4.5 * n_valu / simd_cycles
        print(f"| {k} | {t * 1e3:.2f} | {clock / 1e9:.2f} | {mfma:.2f} | {valu_ctr:.2f} | {mfma + valu_ctr:.2f} | {valu_est:.2f} | "
              f"{mfma + valu_est:.2f} | {n_valu / c['SQ_INSTS_MFMA']:.2f} | {c.get('SQ_INSTS_VALU_TRANS_F32', 0) / c['SQ_INSTS_MFMA']:.3f} |")


if __name__ == '__main__':
    main()
