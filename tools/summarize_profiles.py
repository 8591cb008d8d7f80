"""Turn the rocprofv3 CSVs of a profiled `bench.py` run into the summaries kept under profiles/.

    python tools/summarize_profiles.py gpurun_out/prof_r02 r02      (directory written by tools/run_profiles.sh)

expects <dir>/stats_kernel_stats.csv (--kernel-trace --stats), <dir>/fetch_counter_collection.csv
(--pmc FETCH_SIZE) and <dir>/write_counter_collection.csv (--pmc WRITE_SIZE), each with the bench log
next to it, and writes profiles/<tag>_cfg4_kernel_stats.{md,csv}, profiles/<tag>_cfg4_pmc_traffic.md and
profiles/pmc_traffic.json (read by bench.py for `roofline.traffic`).

HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE, counters in KiB: on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section)."""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# device kernel behind each timed C-ABI call: the plane-format passes (round 5: csrc/edge_mfma_f16x2.hip) where they ran,
# else the fp32-MFMA kernels (csrc/edge_mfma.hip)
KERNEL_CANDIDATES = {'ampconv_fwd_edge': ('fwd_f16x2', 'fwd_mfma'), 'ampconv_bwd_edge_dst': ('bwd_dst_f16x2', 'bwd_dst_mfma'),
                     'ampconv_bwd_edge_src': ('bwd_src_f16x2', 'bwd_src_mfma')}


def pick_kernels(counter_csv):
    names = {r['Kernel_Name'] for r in csv.DictReader(open(counter_csv))}
    out = {}
    for api, cands in KERNEL_CANDIDATES.items():
        out[api] = next(c for c in cands if any(re.search(r'\b' + c + r'(_t4)?<', n) for n in names))
    return out


def bench_line(path):
    for line in open(path):
        if line.startswith('{"metric"'):
            return json.loads(line)
    raise SystemExit(f'no bench line in {path}')


def short(name, n=118):
    # Tensile kernel names carry their types: ..._SB_... = fp32 in / fp32 out, ..._BBS_BH_... = bf16 in, fp32 accumulate
    name = re.sub(r'Cijk_\w+', lambda m: ('rocBLAS/Tensile bf16 GEMM ' if 'BBS' in m.group(0) or '_BB' in m.group(0)
                                          else 'rocBLAS/Tensile fp32 GEMM ') + m.group(0)[:14], name)
    name = name.replace('(anonymous namespace)::', '')
    return name if len(name) <= n else name[:n]


def step_table(trace_csv, first_pat, last_pat):
    """Every kernel launched inside the LAST step of a profiled bench run, from the kernel trace.  A step opens with the
    graph preparation (two `first_pat` = extract_keys launches, one per sort) and closes with the input-gradient
    projection behind the last weight-gradient reduction: the window runs from the first extract_keys of the last pair to
    the last launch matching `last_pat`.  Returns markdown lines (kernel, launches, total ms)."""
    rows = sorted(csv.DictReader(open(trace_csv)), key=lambda r: int(r['Start_Timestamp']))
    names = [r['Kernel_Name'] for r in rows]
    last = max(i for i, n in enumerate(names) if re.search(last_pat, n))
    keys = [i for i, n in enumerate(names[:last]) if re.search(first_pat, n)]
    first = keys[-2] if len(keys) >= 2 else keys[-1]
    agg = {}
    for r in rows[first:last + 1]:
        a = agg.setdefault(r['Kernel_Name'], [0, 0])
        a[0] += 1
        a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    span = (int(rows[last]['End_Timestamp']) - int(rows[first]['Start_Timestamp'])) / 1e6
    out = [f'Window: {span:.1f} ms from the first `extract_keys` launch of the step (graph preparation) to its last projection launch; '
           f'{sum(a[0] for a in agg.values())} launches, {sum(a[1] for a in agg.values()) / 1e6:.1f} ms of kernel time.\n\n',
           '| kernel | launches | total ms |\n|---|---|---|\n']
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append(f'| `{short(n)}` | {c} | {t / 1e6:.3f} |\n')
    return out


def counter(path, device_kernel, launches):
    tot, calls = 0.0, 0
    for row in csv.DictReader(open(path)):
        if re.search(r'\b' + device_kernel + r'(_t4)?<', row['Kernel_Name']):
            tot += float(row['Counter_Value'])
            calls += 1
    assert calls == launches, (device_kernel, calls)
    return tot / calls


def main():
    d, tag = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, 'profiles')
    b = bench_line(os.path.join(d, 'bench_stats.log'))
    cfg = b['config']
    rows = list(csv.DictReader(open(os.path.join(d, 'stats_kernel_stats.csv'))))
    shutil.copy(os.path.join(d, 'stats_kernel_stats.csv'), os.path.join(out, f'{tag}_cfg4_kernel_stats.csv'))
    ms = b['kernels_ms']
    with open(os.path.join(out, f'{tag}_cfg4_kernel_stats.md'), 'w') as f:
        f.write(f'# rocprofv3 --kernel-trace --stats ({tag}, final code of the round)\n\n')
        f.write(f'Command (on the MI355X box, tools/run_profiles.sh): `rocprofv3 --kernel-trace --stats --output-format csv -d {d} '
                '-o stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-gemm --no-extra`\n\n')
        f.write(f"Workload {cfg['workload'].split(':')[0]} (N={cfg['N']}, E={cfg['E']}, L={cfg['L']}, D={cfg['D']}, "
                f"H={cfg['H']}, fp32), 3 steps recorded (1 warm-up + 2 timed).\n")
        f.write(f"Bench line printed by this profiled run: {b['value'] / 1e6:.2f} M edges/s, {b['ms_per_step']:.1f} "
                f"ms/step; HIP-event averages inside bench.py: fwd {ms['ampconv_fwd_edge']:.1f} / bwd_dst "
                f"{ms['ampconv_bwd_edge_dst']:.1f} / bwd_src {ms['ampconv_bwd_edge_src']:.1f} ms.\n\n")
        f.write('`fwd_f16x2` / `bwd_dst_f16x2` / `bwd_src_f16x2` = the plane-format edge passes (csrc/edge_mfma_f16x2.hip); '
                '`proj_rows_kernel<..., true, true>` = the two projections that leave as fp16 planes (Q|K|V, dObar), `<..., true, false>` '
                'the two with fp32 output (out-projection, dX), `proj_wgrad_kernel` the two weight-gradient products (csrc/proj_gemm.hip); '
                '`absmax_stats_kernel` = the range / maximum pass over dY (x: once, cached); `__amd_rocclr_copyBuffer` and `distribution_elementwise...` are the '
                "bench's own data fill (make_batch: 1 GiB slabs of randn copied into x and dy), outside the timed steps.\n\n")
        f.write('| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|\n')
        for r in rows[:22]:
            f.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.3f} | "
                    f"{float(r['TotalDurationNs']) / 1e6:.1f} | {r['Percentage']} |\n")
        tr = os.path.join(d, 'stats_kernel_trace.csv')
        if os.path.exists(tr):
            f.write('\nEverything launched inside ONE timed step (kernel trace of the same run):\n\n')
            f.writelines(step_table(tr, r'extract_keys', r'proj_rows_kernel|wgrad_reduce_kernel'))
    N, E, L, D = cfg['N'], cfg['E'], cfg['L'], cfg['D']
    R = L * D * 4
    alg = {'ampconv_fwd_edge': (2 * E + 2 * N) * R, 'ampconv_bwd_edge_dst': (2 * E + 3 * N) * R,
           'ampconv_bwd_edge_src': (2 * E + 4 * N) * R}
    stats_bytes = E * cfg['H'] * 40 * 4
    js = {'workload': 'cfg4', 'round': tag,
          'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (KiB units); FETCH_SIZE doubled '
                  'per MI355X_MICROARCH.md (gfx950 reports half of 16-B/lane coalesced reads); per launch',
          'kernels': {}}
    lines = []
    for api, dk in pick_kernels(os.path.join(d, 'fetch_counter_collection.csv')).items():
        fetch = counter(os.path.join(d, 'fetch_counter_collection.csv'), dk, 2)
        write = counter(os.path.join(d, 'write_counter_collection.csv'), dk, 2)
        traffic = (2 * fetch + write) * 1024
        js['kernels'][api] = {'device_kernel': dk, 'fetch_size_kib_raw': fetch, 'write_size_kib': write,
                              'traffic_bytes': traffic, 'algorithmic_bytes': alg[api]}
        extra = 0 if api == 'ampconv_fwd_edge' else stats_bytes
        lines.append(f'| `{dk}` | {fetch:.0f} | {write:.0f} | {traffic / 1e9:.1f} GB | {alg[api] / 1e9:.1f} GB | '
                     f'{extra / 1e9:.1f} GB | {traffic / (alg[api] + extra):.3f} |\n')
    json.dump(js, open(os.path.join(out, 'pmc_traffic.json'), 'w'), indent=1)
    with open(os.path.join(out, f'{tag}_cfg4_pmc_traffic.md'), 'w') as f:
        f.write(f'# rocprofv3 PMC HBM traffic ({tag}, final code), cfg4 (1M nodes / 10M edges, L=20, D=256, H=8)\n\n')
        f.write(f'Commands: `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d {d} -o fetch -- '
                'python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-gemm` and the same with `--pmc WRITE_SIZE '
                '-o write` (separate passes: TCC slots).\n\nUnits: counter values are KiB; FETCH_SIZE is doubled (gfx950 '
                'counts 128-B requests of wide coalesced reads as 64 B, MI355X_MICROARCH.md "HBM").  "statistics" = the '
                'softmax statistics the dst pass writes and the src pass reads (160 B per edge and head), which are not '
                'part of the algorithmic byte count of SURVEY.md 8d.\n\n')
        f.write('| kernel | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM bytes per launch (2*fetch + write) | algorithmic '
                'bytes per launch | statistics | traffic / (algorithmic + statistics) |\n|---|---|---|---|---|---|---|\n')
        f.writelines(lines)
    # cfg5 (RMAT, bf16 storage) kernel stats of the same session, if present
    c5 = os.path.join(d, 'cfg5_kernel_stats.csv')
    if os.path.exists(c5):
        b5 = bench_line(os.path.join(d, 'bench_cfg5.log'))
        shutil.copy(c5, os.path.join(out, f'{tag}_cfg5_kernel_stats.csv'))
        rows5 = list(csv.DictReader(open(c5)))
        with open(os.path.join(out, f'{tag}_cfg5_kernel_stats.md'), 'w') as f:
            f.write(f'# rocprofv3 --kernel-trace --stats, cfg5 ({tag}, final code of the round)\n\n')
            f.write(f'Command: `rocprofv3 --kernel-trace --stats --output-format csv -d {d} -o cfg5 -- python3 bench.py '
                    '--workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline`\n\n')
            c = b5['config']
            k5 = b5['kernels_ms']
            f.write(f"Workload {c['workload'].split(',')[0]} (N={c['N']}, E={c['E']}, L={c['L']}, D={c['D']}, H={c['H']}, bf16 "
                    f"storage), 3 steps recorded.  Bench line of this profiled run: {b5['value'] / 1e6:.2f} M edges/s, "
                    f"{b5['ms_per_step']:.1f} ms/step; HIP-event averages per C-ABI call (main + long-segment pass + combine): "
                    f"fwd {k5['ampconv_fwd_edge']:.1f} / bwd_dst {k5['ampconv_bwd_edge_dst']:.1f} / bwd_src "
                    f"{k5['ampconv_bwd_edge_src']:.1f} ms.\n\n")
            f.write('| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|\n')
            for r in rows5[:22]:
                f.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.3f} | "
                        f"{float(r['TotalDurationNs']) / 1e6:.1f} | {r['Percentage']} |\n")
            f.write('\nThe `at::native::*` rows above are the bench\'s set-up (R-MAT generator, randn + cast of the inputs), '
                    'not the layer.  Everything launched inside ONE timed step, from the kernel trace of the same run '
                    '(`cfg5_kernel_trace.csv`): no `Cijk_*` (rocBLAS / Tensile) and no `at::native::*` kernel.\n\n')
            f.writelines(step_table(os.path.join(d, 'cfg5_kernel_trace.csv'), r'extract_keys', r'proj_rows_bf16_kernel|mask_rows_kernel|wgrad_reduce_bf16'))
    # GraphSAINT batches of config 4, if profiled
    cs = os.path.join(d, 'saint_kernel_stats.csv')
    if os.path.exists(cs) and os.path.exists(os.path.join(d, 'bench_saint.log')):
        b = bench_line(os.path.join(d, 'bench_saint.log'))
        shutil.copy(cs, os.path.join(out, f'{tag}_cfg4_saint_kernel_stats.csv'))
        rows_s = list(csv.DictReader(open(cs)))
        nb = b['steps'] + b['warmup']
        with open(os.path.join(out, f'{tag}_cfg4_saint_kernel_stats.md'), 'w') as f:
            f.write(f'# rocprofv3 --kernel-trace --stats, GraphSAINT batches of config 4 ({tag}, final code of the round)\n\n')
            f.write(f'Command: `rocprofv3 --kernel-trace --stats --output-format csv -d {d} -o saint -- python3 bench.py '
                    '--workload cfg4-saint --steps 10 --warmup 2 --no-cpu-baseline`\n\n')
            f.write(f"Bench line of this profiled run: {b['value'] / 1e6:.2f} M sampled edges/s, {b['ms_per_step']:.2f} ms per batch "
                    f"({b['nodes_avg']:.0f} nodes / {b['edges_avg']:.0f} edges per batch, sampler {b['sampler_ms']:.2f} ms); {nb} "
                    f"batches recorded ({b['warmup']} warm-up + {b['steps']} timed).  The `distribution_elementwise` / `copyBuffer` rows "
                    "are the bench's data fill of the resident 1 M-node graph, outside the timed batches.\n\n")
            f.write('| kernel | calls | calls per batch | avg ms | total ms | % |\n|---|---|---|---|---|---|\n')
            for r in rows_s[:24]:
                f.write(f"| `{short(r['Name'], 110)}` | {r['Calls']} | {int(r['Calls']) / nb:.1f} | {float(r['AverageNs']) / 1e6:.3f} | "
                        f"{float(r['TotalDurationNs']) / 1e6:.1f} | {r['Percentage']} |\n")
    sq = os.path.join(d, 'sq_summary.txt')
    if os.path.exists(sq):
        with open(os.path.join(out, f'{tag}_sq_counters.md'), 'w') as f:
            f.write(f'# SQ counters of the edge and projection kernels ({tag}; tools/prof_sq.sh, tools/prof_fp32_pipe.sh, tools/prof_sq_proj.sh)\n\n')
            f.write('Collected by `tools/run_profiles.sh <tag> counters` on the final code of the round.  Per kernel family: two '
                    '`rocprofv3 --pmc ... --kernel-trace` passes (`tools/prof_sq.sh` over `tools/bench_kernels.py`, '
                    '`tools/prof_sq_proj.sh` over `tools/bench_proj.py`): `SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY '
                    'SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT '
                    'GRBM_GUI_ACTIVE` and `SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS '
                    'SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE`; fractions are of SQ_WAVE_CYCLES; mfma_busy = '
                    'SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (duration x clock), clock = GRBM_GUI_ACTIVE / 8 / duration.  '
                    'The `pipe` tables (`tools/prof_fp32_pipe.sh`, one more pass: `SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU '
                    'SQ_INSTS_VALU_TRANS_F32` added) are the FP32-pipe occupancy of VERDICT r3 item 2: matrix pipe busy + VALU '
                    'active over SIMD cycles (fp32-input MFMA and VALU share one pipe on gfx950: `tools/coexec_probe`); for the '
                    'bf16 kernels, whose MFMAs co-issue with VALU, the two columns are separate resources and the VALU column is '
                    'the binding one.  Shapes: fp32 edge kernels 100 k nodes / 1 M edges, L=20, D=256, H=8 (uniform); bf16 edge '
                    'kernels R-MAT scale 19 / 10 M edges (config 5 at a quarter of its size); workgroup-per-unit kernels L=40, D=100, H=2 (the '
                    'reference\'s AMPGCN defaults; since round 5 csrc/edge_block_x3.hip: `*_xh` = two fp16 planes behind the bound-carrying '
                    'entry points, `*_x3` = three bf16 planes behind the fp32 ones; profiles/r05_x3_sq.md has their HBM traffic); projections 8 M rows, D = 256 (`proj_rows` averages the qkv / out / dx launches, '
                    '`proj_wgrad` the dWin / dWo launches).  The projections\' clock column is the DVFS evidence of DESIGN.md 4a: '
                    '1.54-1.75 GHz on random data in both the six-product and the scaled three-product form.  Profiled runs are a '
                    'few % slower than un-profiled ones.\n\n```\n')
            f.write(open(sq).read())
            f.write('```\n')


if __name__ == '__main__':
    main()
