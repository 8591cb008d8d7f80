"""Turn the rocprofv3 CSVs of a profiled `bench.py` run into the summaries kept under profiles/.

    python tools/summarize_profiles.py gpurun_out/prof2 r01

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
KERNELS = {'ampconv_fwd_edge': 'fwd_mfma', 'ampconv_bwd_edge_dst': 'bwd_dst_mfma',
           'ampconv_bwd_edge_src': 'bwd_src_mfma'}


def bench_line(path):
    for line in open(path):
        if line.startswith('{"metric"'):
            return json.loads(line)
    raise SystemExit(f'no bench line in {path}')


def short(name, n=118):
    name = re.sub(r'Cijk_\w+', lambda m: 'rocBLAS/Tensile fp32 GEMM ' + m.group(0)[:14], name)
    return name if len(name) <= n else name[:n]


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
        f.write('Command (on the MI355X box): `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2 '
                '-o stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-gemm`\n\n')
        f.write(f"Workload {cfg['workload'].split(':')[0]} (N={cfg['N']}, E={cfg['E']}, L={cfg['L']}, D={cfg['D']}, "
                f"H={cfg['H']}, fp32), 3 steps recorded (1 warm-up + 2 timed).\n")
        f.write(f"Bench line printed by this profiled run: {b['value'] / 1e6:.2f} M edges/s, {b['ms_per_step']:.1f} "
                f"ms/step; HIP-event averages inside bench.py: fwd {ms['ampconv_fwd_edge']:.1f} / bwd_dst "
                f"{ms['ampconv_bwd_edge_dst']:.1f} / bwd_src {ms['ampconv_bwd_edge_src']:.1f} ms.\n\n")
        f.write('| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|\n')
        for r in rows[:22]:
            f.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.3f} | "
                    f"{float(r['TotalDurationNs']) / 1e6:.1f} | {r['Percentage']} |\n")
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
    for api, dk in KERNELS.items():
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
        f.write('Commands: `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof2 -o fetch -- '
                'python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-gemm` and the same with `--pmc WRITE_SIZE '
                '-o write` (separate passes: TCC slots).\n\nUnits: counter values are KiB; FETCH_SIZE is doubled (gfx950 '
                'counts 128-B requests of wide coalesced reads as 64 B, MI355X_MICROARCH.md "HBM").  "statistics" = the '
                'softmax statistics the dst pass writes and the src pass reads (160 B per edge and head), which are not '
                'part of the algorithmic byte count of SURVEY.md 8d.\n\n')
        f.write('| kernel | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM bytes per launch (2*fetch + write) | algorithmic '
                'bytes per launch | statistics | traffic / (algorithmic + statistics) |\n|---|---|---|---|---|---|---|\n')
        f.writelines(lines)


if __name__ == '__main__':
    main()
