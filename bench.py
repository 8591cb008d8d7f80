#!/usr/bin/env python3
"""bench.py -- AMPConv edges/sec (forward + backward) on MI355X.

One "step" = one pass of the hot path over one batch: CSR/CSC build of the
batch's edge_index + one AMPConv layer forward + backward (gradients w.r.t. x
and the four parameters), fp32, on a synthetic uniform random graph of the
shape BASELINE.json names (config 4: 1 M nodes / 10 M edges, L=20 tokens,
D=256, 8 heads).  With N > 1 GPUs every rank processes its OWN graph of that
shape (one-subgraph-per-GPU data parallelism, weak scaling) and the parameter
gradients are averaged with one RCCL all-reduce per step inside the timed
region (experiments/cora_benchmark_graphsaint_distributed.py:63-94 as intended).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg4|cfg3|cfg3-L1|cfg3-L4|cfg5|cora|ampgcn-default|cfg4-saint|tiny] [--dtype f32|bf16] [--gemm native|fp32|bf16x3]

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel against
the 8 TB/s HBM3E spec peak with durations measured live by HIP events on the
launch stream over the timed region; `cpu_baseline` is the reference-shaped CPU
restatement (oracle/ampconv_torch.py) timed on the host cores on a bounded
sub-sample of the same workload (rank 0, N=1 only).  Beside the headline the line
carries: `saint` (every N: BASELINE config 4 as written -- the resident graph, one
GraphSAINT subgraph per rank per step -- measured on the same ranks right after
the full-graph pass), `extra_workloads` (N = 1: configs 3 and 5, the L = 1 / L = 4
sweeps of config 3, the Cora-sized config 1/2 shape and the reference's class-default
layer shape, a few steps each), `alt_gemm` (the same step on the library's fp32 GEMMs),
`dist` (backend, world size and the number of ranks an all-reduce of ones saw) and
per-rank kernel and all-reduce times, so that one line diagnoses a scaling run.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # name: (N, E, L, D, H, description)
    'cfg4': (1_000_000, 10_000_000, 20, 256, 8,
             'BASELINE config 4: uniform random graph 1M nodes / 10M edges, L=20, D=256, H=8'),
    'cfg3': (100_000, 1_000_000, 20, 128, 8,
             'BASELINE config 3: uniform random graph 100k nodes / 1M edges, L=20, D=128, H=8'),
    'cfg5': (1 << 21, 40_000_000, 20, 256, 8,
             'BASELINE config 5: RMAT (0.57,0.19,0.19,0.05) scale 21 = 2.1M nodes / 40M edges, L=20, D=256, H=8, '
             'bf16 storage'),
    'cora': (2708, 10556, 20, 128, 4, 'BASELINE config 1/2 shape: Cora-sized random graph, L=20, D=128, H=4'),
    # the L sweeps of config 3 (BASELINE.md section 4 has a roofline row for L = 1)
    'cfg3-L1': (100_000, 1_000_000, 1, 128, 8, 'BASELINE config 3, L = 1 sweep: 100k nodes / 1M edges, L=1, D=128, H=8'),
    'cfg3-L4': (100_000, 1_000_000, 4, 128, 8, 'BASELINE config 3, L = 4 sweep: 100k nodes / 1M edges, L=4, D=128, H=8'),
    # the shape the reference's own scripts instantiate: AMPGCN class defaults (src/ampnet/module/amp_gcn.py:21-35,
    # experiments/cora_benchmark_graphsaint_distributed.py:58, cora_benchmark_full.py:46) on a config-3-sized graph
    'ampgcn-default': (100_000, 1_000_000, 40, 100, 2,
                       'AMPGCN class-default layer shape (L=40, D=100, H=2: amp_gcn.py:21-35) on 100k nodes / 1M edges'),
    'tiny': (2000, 20000, 20, 256, 8, 'smoke-sized graph of the config-4 layer shape'),
    # GraphSAINT batches (BASELINE config 4 as written): the graph and its features stay resident, every step
    # draws one random-walk subgraph per rank (experiments/cora_benchmark_graphsaint.py:80-82,96-116)
    'cfg4-saint': (1_000_000, 10_000_000, 20, 256, 8,
                   'BASELINE config 4, GraphSAINT batches: resident uniform random graph 1M nodes / 10M edges, '
                   'per step one random-walk subgraph per GPU (2000 roots x walk length 50), L=20, D=256, H=8'),
    'tiny-saint': (20_000, 200_000, 20, 256, 8, 'smoke-sized GraphSAINT-batch run (200 roots x walk length 10)'),
}
SAINT = {'cfg4-saint': (2000, 50), 'tiny-saint': (200, 10)}          # (roots per batch, walk length)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {'f32': 157.3, 'bf16': 2500.0}   # same guide: fp32-input MFMA = vector rate; dense bf16


class KernelTimer:
    """HIP events (torch.cuda.Event on the launch stream = torch's current stream) around
    each C-ABI edge kernel; installed into ampnet_amd.conv.functional for the timed steps."""

    def __init__(self):
        self.events = {}
        self.entry_points = set()
        self.enabled = False

    def wrap(self, lib, name, key=None):
        """Events around C entry point `name`, filed under `key` (the plane-format edge passes of ABI 106 and the
        bound-carrying fp32-view passes of ABI 107 are filed under the name of the fp32 entry point whose work they do;
        `entry_points` keeps what really ran)."""
        fn = getattr(lib, name)
        key = key or name

        def timed(*args):
            if not self.enabled:
                return fn(*args)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            self.events.setdefault(key, []).append((e0, e1))
            self.entry_points.add(name)
            return rc
        return timed

    def summary(self):
        return {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in self.events.items() if v}


class TimedLib:
    def __init__(self, lib, timer):
        self._lib = lib
        for n in ('ampconv_fwd_edge', 'ampconv_bwd_edge_dst', 'ampconv_bwd_edge_src'):
            setattr(self, n, timer.wrap(lib, n))
            setattr(self, n + '_planes', timer.wrap(lib, n + '_planes', key=n))
            setattr(self, n + '_scaled', timer.wrap(lib, n + '_scaled', key=n))

    def __getattr__(self, name):
        return getattr(self._lib, name)


def rmat_edges(scale, E, gen, dev, a=0.57, b=0.19, c=0.19):
    """R-MAT edge list (no de-duplication): per bit one quadrant draw with probabilities a, b, c, d."""
    src = torch.zeros(E, dtype=torch.int64, device=dev)
    dst = torch.zeros(E, dtype=torch.int64, device=dev)
    for _ in range(scale):
        r = torch.rand(E, generator=gen, device=dev)
        src = src * 2 + (r >= a + b).to(torch.int64)
        dst = dst * 2 + (((r >= a) & (r < a + b)) | (r >= a + b + c)).to(torch.int64)
    return torch.stack([src, dst])


def make_batch(N, E, L, D, seed, dev, dtype=torch.float32, rmat=False):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.empty(N, L * D, device=dev, dtype=dtype)
    dy = torch.empty(N, L * D, device=dev, dtype=dtype)
    rows = max(1, (1 << 28) // (L * D))                      # fill in slabs: no full-size fp32 temporary
    for t in (x, dy):
        for r0 in range(0, N, rows):
            t[r0:r0 + rows] = torch.randn(min(rows, N - r0), L * D, generator=g, device=dev).to(dtype)
    if rmat:
        ei = rmat_edges(N.bit_length() - 1, E, g, dev)
    else:
        ei = torch.randint(0, N, (2, E), generator=g, device=dev, dtype=torch.int64)
    return x, ei, dy


def host_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get('AMPCONV_CPU_THREADS', n))))


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(L, D, H, ratio, budget_s=20.0):
    """Reference-shaped CPU path (gather -> nn.MultiheadAttention -> scatter-mean), fwd+bwd, on a bounded
    sub-sample with the workload's L, D, H and E/N (1 warm-up + 3 timed, best; about 100 k edges when
    the host allows).  The sample is sized from a small timed probe so that the three timed passes take about
    `budget_s` seconds, and capped at 100 k edges and a quarter of the host RAM (~1 KB per edge and channel)."""
    from oracle.ampconv_torch import RefShapedAMPConv
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(1)
    layer = RefShapedAMPConv(D, H)
    g = torch.Generator().manual_seed(2)

    def sample(n_cpu):
        e_cpu = int(n_cpu * ratio)
        x = torch.randn(n_cpu, L * D, generator=g, requires_grad=True)
        return x, torch.randint(0, n_cpu, (2, e_cpu), generator=g), torch.randn(n_cpu, L * D, generator=g), e_cpu

    def step(x, edges, dy):
        layer.zero_grad(set_to_none=True)
        x.grad = None
        y = layer(x, edges)
        (y * dy).sum().backward()

    x, ei, dy, e_probe = sample(400)                       # probe (also warms the thread pool and the allocator)
    step(x, ei, dy)
    t0 = time.perf_counter()
    step(x, ei, dy)
    rate = e_probe / (time.perf_counter() - t0)
    try:
        ram = os.sysconf('SC_PAGE_SIZE') * os.sysconf('SC_PHYS_PAGES')
    except (ValueError, OSError):
        ram = 64 << 30
    e_cpu = int(min(100_000, rate * budget_s / 3, 0.25 * ram / (1000.0 * D)))
    n_cpu = max(400, int(e_cpu / ratio))
    x, ei, dy, e_cpu = sample(n_cpu)
    step(x, ei[:, : max(1, e_cpu // 8)], dy)               # warm-up on an eighth of the sample
    best = float('inf')
    for _ in range(3):
        t0 = time.perf_counter()
        step(x, ei, dy)
        best = min(best, time.perf_counter() - t0)
    return {'value': e_cpu / best, 'unit': 'edges/s', 'cores': cores, 'kind': 'port', 'cpu_model': cpu_model(),
            'sample': f'{n_cpu} nodes / {e_cpu} edges, L={L} D={D} H={H} fp32, fwd+bwd, 1 warm-up + 3 timed (best), '
                      f'torch {torch.__version__} CPU, {cores} threads; sample sized for ~{budget_s:.0f} s of CPU work '
                      f'(the reference materialises ~{D} KB per edge)'}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default=os.environ.get('AMPCONV_BENCH_WORKLOAD', 'cfg4'))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt-gemm', action='store_true',
                    help='skip the extra pass with the projections on the library GEMMs (profiling runs)')
    ap.add_argument('--no-extra', action='store_true',
                    help='skip the side measurements of the default run (N = 1: configs 3 and 5; N > 1: GraphSAINT batches)')
    ap.add_argument('--softmax-free', action='store_true',
                    help="the reference's softmax-free attention variant (AMPConv(..., softmax=False)); not the headline")
    ap.add_argument('--dtype', default=None, choices=['f32', 'bf16'],
                    help='storage dtype (default: f32; bf16 for cfg5)')
    ap.add_argument('--gemm', default='native', choices=['native', 'fp32', 'bf16x3'],
                    help="per-node projections: 'native' (default, the headline: libampconv's own kernels, fp32 operands "
                         "scaled by a power of two and split into two fp16 planes, three partial products on the matrix "
                         "cores -- six bf16 ones for small operands --, error below the fp32 library GEMM's "
                         "-- csrc/proj_gemm.hip), 'fp32' (library GEMMs, rocBLAS) or 'bf16x3' (hipBLASLt's "
                         "3-product split, 8x the error; ampnet_amd.conv.functional.gemm_precision)")
    return ap.parse_args()


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: start N fresh worker processes of this
    script (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) BEFORE this
    process touches the GPU and watch them: the first rank that exits non-zero ends the run -- the others
    (blocked in a collective that will never complete) are terminated, then killed after a grace period --
    and (rank, exit code) of every failed rank is reported.  Mirrors the reference's own mp.spawn of one
    process per rank (experiments/cora_benchmark_graphsaint_distributed.py:130-142).  Rank 0 prints the JSON line."""
    import socket
    import subprocess
    port = os.environ.get('MASTER_PORT')
    if port is None:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = str(s.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [None] * n
    first_bad = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
                if rcs[r] not in (None, 0) and first_bad is None:
                    first_bad = r
        if first_bad is not None:
            break
        time.sleep(0.05)
    if first_bad is not None:
        survivors = [r for r in range(n) if rcs[r] is None]
        for r in survivors:
            procs[r].terminate()
        deadline = time.time() + 10.0
        for r in survivors:
            try:
                rcs[r] = procs[r].wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                rcs[r] = procs[r].wait()
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0 and r not in survivors]
        print(f'bench.py: ranks failed (rank, exit code): {bad}; first: rank {first_bad}; '
              f'stopped the {len(survivors)} rank(s) still running: {survivors}', file=sys.stderr)
        sys.exit(1)
    sys.exit(0)


class _stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when its first communicator comes up; the contract is ONE JSON line there.
    File-descriptor level redirect (the banner comes from C code) for the duration of the block."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def _fence(dist_on):
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()


def _gather_ranks(values, dev, dist_on):
    """[len(values)] floats of every rank -> list (by rank) of lists, on every rank."""
    if not dist_on:
        return [list(values)]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, [float(v) for v in values])
    return out


EDGE_CALLS = ('ampconv_fwd_edge', 'ampconv_bwd_edge_dst', 'ampconv_bwd_edge_src')


def _edge_alg_bytes(N, E, R):
    return {'ampconv_fwd_edge': (2 * E + 2 * N) * R, 'ampconv_bwd_edge_dst': (2 * E + 3 * N) * R,
            'ampconv_bwd_edge_src': (2 * E + 4 * N) * R}


def _per_rank(ms_by_rank, dist_on):
    return [{'rank': r, **{k.replace('ampconv_', ''): v[i] for i, k in enumerate(EDGE_CALLS)},
             **({'allreduce': v[3]} if dist_on else {})} for r, v in enumerate(ms_by_rank)]


def make_layer(D, H, tdt, dev, args, dist_on, softmax=True):
    from ampnet_amd import AMPConv
    from ampnet_amd.distributed import GradientAllReducer, broadcast_parameters
    torch.manual_seed(1)
    layer = AMPConv(D, H, softmax=softmax).to(dev)
    layer.retain_attention = False                            # no [E, L, L] side output kept alive
    layer.gemm_precision = args.gemm
    with torch.no_grad():
        layer.multi_head_attention.in_proj_bias.normal_(0, 0.1)
        layer.multi_head_attention.out_proj.bias.normal_(0, 0.1)
    layer = layer.to(tdt)
    if dist_on:
        broadcast_parameters(layer, src=0)
    return layer, GradientAllReducer(layer.parameters())


def measure_saint(workload, steps, warmup, args, rank, world, dev, dist_on, dt_name):
    """GraphSAINT-batch mode (BASELINE config 4 as written; experiments/cora_benchmark_graphsaint_distributed.py:64-94):
    per rank per step GraphSAINTRandomWalkSampler.sample() -> row gather of the resident features -> AMPConv forward +
    backward on the induced subgraph -> gradient all-reduce.  Returns (on every rank) the measurement as a dict:
    `value` = sampled edges processed by all ranks per second; the sampler's and the all-reduce's share of the step,
    the host read-backs, per-rank kernel times."""
    import types
    from ampnet_amd import GraphSAINTRandomWalkSampler, graph_cache, _lib
    from ampnet_amd.sampler import gather_rows
    N, E, L, D, H, desc = WORKLOADS[workload]
    tdt = torch.bfloat16 if dt_name == 'bf16' else torch.float32
    R = L * D * (2 if dt_name == 'bf16' else 4)
    layer, reducer = make_layer(D, H, tdt, dev, args, dist_on)
    roots, walk = SAINT[workload]
    x, ei, _ = make_batch(N, E, L, D, 13, dev, tdt)           # the same resident graph on every rank (seed 13)
    del _
    sampler = GraphSAINTRandomWalkSampler(types.SimpleNamespace(edge_index=ei, num_nodes=N), batch_size=roots,
                                          walk_length=walk, num_steps=1, seed=13 + rank, num_nodes=N)
    n_max = roots * (walk + 1)
    g = torch.Generator(device=dev).manual_seed(99 + rank)
    dy_buf = torch.randn(n_max, L * D, generator=g, device=dev).to(tdt)

    timer = KernelTimer()
    timed_lib = TimedLib(_lib.load(), timer)
    orig_load = _lib.load
    _lib.load = lambda: timed_lib
    ev, ar_ev, sizes = [], [], []

    def step(timed):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        node_idx, ei_sub, _, _ = sampler.sample()
        e1.record()
        graph_cache.clear()
        layer.zero_grad(set_to_none=True)
        xs = gather_rows(x, node_idx).requires_grad_(True)       # (the library's gather kernel: csrc/sampler.hip)
        y = layer(xs, ei_sub)
        y.backward(dy_buf[: xs.size(0)])
        if dist_on:
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            reducer.allreduce()
            a1.record()
            if timed:
                ar_ev.append((a0, a1))
        if timed:
            ev.append((e0, e1))
            sizes.append((int(node_idx.numel()), int(ei_sub.size(1))))

    try:
        for _ in range(warmup):
            step(False)
        _fence(dist_on)
        timer.enabled = True
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        _fence(dist_on)
        dt = time.perf_counter() - t0
    finally:
        timer.enabled = False
        _lib.load = orig_load
    tot = torch.tensor([dt, float(sum(e for _, e in sizes)), float(sum(n for n, _ in sizes))], dtype=torch.float64,
                       device=dev)
    if dist_on:
        tmax = tot[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        tot[0] = tmax[0]
    dt, edges_all, nodes_all = (float(v) for v in tot.tolist())
    ms = timer.summary()
    ar_ms = sum(a.elapsed_time(b) for a, b in ar_ev) / len(ar_ev) if ar_ev else 0.0
    by_rank = _gather_ranks([ms.get(k, 0.0) for k in EDGE_CALLS] + [ar_ms], dev, dist_on)
    n_avg = nodes_all / (steps * world)
    e_avg = edges_all / (steps * world)
    alg = _edge_alg_bytes(sum(n for n, _ in sizes) / len(sizes), sum(e for _, e in sizes) / len(sizes), R)
    dom = max(ms, key=ms.get)
    achieved = alg[dom] / (ms[dom] * 1e-3) / 1e9
    sampler_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    flops = 14 * L * L * D * e_avg + 24 * L * D * D * n_avg
    t_step = dt / steps
    del x, ei, dy_buf, sampler
    graph_cache.clear()
    torch.cuda.empty_cache()
    return {
        'metric': 'AMPConv edges/sec (fwd+bwd), GraphSAINT subgraph batches', 'value': edges_all / dt, 'unit': 'edges/s',
        'n_gpus': world, 'steps': steps, 'warmup': warmup, 'ms_per_step': 1e3 * t_step,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dt_name, 'data': 'synthetic',
        'config': {'workload': f'{workload}: {desc}; sampler + feature-row gather + one AMPConv layer fwd+bwd incl. CSR '
                               f'build of the subgraph' + (' + RCCL grad all-reduce' if world > 1 else ''),
                   'N': N, 'E': E, 'L': L, 'D': D, 'H': H, 'roots': roots, 'walk_length': walk,
                   'parallelism': f'dp{world}', 'gemm': args.gemm if dt_name == 'f32' else 'bf16',
                   'projections': _projection_note(dt_name, args.gemm),
                   'edge_entry_points': sorted(timer.entry_points)},
        'nodes_avg': n_avg, 'edges_avg': e_avg, 'sampler_ms': sampler_ms, 'allreduce_ms': ar_ms,
        'batch': {'nodes_avg': n_avg, 'edges_avg': e_avg, 'sampler_ms': sampler_ms,
                  'sampler_share_of_step': sampler_ms / (1e3 * t_step),
                  'allreduce_share_of_step': ar_ms / (1e3 * t_step),
                  'host_readbacks_per_batch': 'sampler: sub-graph node and edge counts (1 read of both), graph preparation: '
                                              'bounds flag + long-segment chunk counts (1 read of the status word); '
                                              'all inside the timed step'},
        'layer_flops': {'per_step': flops, 'achieved_TFLOPs_per_gpu': flops / t_step / 1e12,
                        'peak_TFLOPs': MFMA_PEAK_TFLOPS[dt_name],
                        'frac_of_peak': flops / t_step / 1e12 / MFMA_PEAK_TFLOPS[dt_name]},
        'kernels_ms': ms, 'per_rank_ms': _per_rank(by_rank, dist_on),
        'roofline': {'kernel': dom, 'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': None, 'traffic_source': None,
                     'algorithmic_bytes_per_launch': alg[dom], 'avg_launch_ms': ms[dom]},
    }


def measure_full(workload, steps, warmup, args, rank, world, dev, dist_on, dt_name, alt=False):
    """One graph of the workload's shape per rank; a step = CSR/CSC build + one AMPConv layer forward + backward
    (+ the gradient all-reduce when a process group is up).  Returns the measurement as a dict (on every rank)."""
    from ampnet_amd import graph_cache, _lib
    N, E, L, D, H, desc = WORKLOADS[workload]
    tdt = torch.bfloat16 if dt_name == 'bf16' else torch.float32
    R = L * D * (2 if dt_name == 'bf16' else 4)
    layer, reducer = make_layer(D, H, tdt, dev, args, dist_on, softmax=not args.softmax_free)
    x, ei, dy = make_batch(N, E, L, D, 1234 + rank, dev, tdt, rmat=workload == 'cfg5')   # own graph per rank
    x.requires_grad_(True)

    timer = KernelTimer()
    timed_lib = TimedLib(_lib.load(), timer)
    orig_load = _lib.load
    _lib.load = lambda: timed_lib                             # functional.py resolves the lib per call
    ar_ev = []

    def step(timed=False):
        graph_cache.clear()                                   # a new batch -> CSR build is in the step
        layer.zero_grad(set_to_none=True)
        x.grad = None
        y = layer(x, ei)
        y.backward(dy)
        if dist_on:
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            reducer.allreduce()
            a1.record()
            if timed:
                ar_ev.append((a0, a1))

    try:
        for _ in range(warmup):
            step()
        _fence(dist_on)
        timer.enabled = True
        step_events = []
        t0 = time.perf_counter()
        for _ in range(steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            step(True)
            e1.record()
            step_events.append((e0, e1))
        _fence(dist_on)
        dt = time.perf_counter() - t0
        timer.enabled = False
        step_ms = sorted(a.elapsed_time(b) for a, b in step_events)
        median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if dist_on:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

        alt_out = None
        if alt:
            # same step with the projections on the library's fp32 GEMMs (rocBLAS) -- what round 2 measured; reported beside
            layer.gemm_precision = 'fp32'
            step()
            _fence(dist_on)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            _fence(dist_on)
            dta = time.perf_counter() - t0
            layer.gemm_precision = args.gemm
            alt_out = {'gemm': 'fp32 (rocBLAS library GEMMs)', 'value': E * steps / dta, 'unit': 'edges/s',
                       'ms_per_step': 1e3 * dta / steps}
    finally:
        timer.enabled = False
        _lib.load = orig_load
    ms = timer.summary()
    ar_ms = sum(a.elapsed_time(b) for a, b in ar_ev) / len(ar_ev) if ar_ev else 0.0
    by_rank = _gather_ranks([ms.get(k, 0.0) for k in EDGE_CALLS] + [ar_ms], dev, dist_on)
    # node lists (bf16 storage, graphs with many edge-less nodes): what the layer's projections actually covered
    used = next((getattr(c, '_active', None) for _, _, c in graph_cache.items), None)
    lists = {k: v[1] for k, v in used.items()} if used else None
    del x, dy, ei
    graph_cache.clear()
    torch.cuda.empty_cache()
    out = {'metric': 'AMPConv edges/sec (fwd+bwd)', 'value': world * E * steps / dt, 'unit': 'edges/s',
           'n_gpus': world, 'steps': steps, 'warmup': warmup, 'ms_per_step': 1e3 * dt / steps, 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': dt_name, 'data': 'synthetic',
           'config': {'workload': f'{workload}: {desc}, one AMPConv layer fwd+bwd incl. CSR build, '
                                  f'one graph per GPU' + (' + RCCL grad all-reduce' if world > 1 else ''),
                      'N': N, 'E': E, 'L': L, 'D': D, 'H': H, 'parallelism': f'dp{world}',
                      'gemm': args.gemm if dt_name == 'f32' else 'bf16',
                      'projections': _projection_note(dt_name, args.gemm),
                      # which C entry points the timed edge calls were (kernels_ms files the plane-format passes of ABI
                      # 106 -- fp16 planes, 16-bit matrix pipe -- under the fp32 entry point whose work they do)
                      'edge_entry_points': sorted(timer.entry_points)}}
    if args.softmax_free:
        # the edge phase of this variant is one segment reduction (conv/linear.py): no edge-kernel roofline
        out['metric'] += ', softmax-free variant'
        return out
    alg = _edge_alg_bytes(N, E, R)
    kflops = {'ampconv_fwd_edge': 4 * L * L * D * E, 'ampconv_bwd_edge_dst': 6 * L * L * D * E,
              'ampconv_bwd_edge_src': 8 * L * L * D * E}   # multiply-adds x 2 of the products each pass does
    dom = max(ms, key=ms.get)
    achieved = alg[dom] / (ms[dom] * 1e-3) / 1e9
    b_alg = (6 * E + 15 * N) * R + 16 * E                 # whole layer, SURVEY.md 8d
    # SURVEY.md 8d names two rooflines: HBM for the edge phase, the matrix pipe for the per-node
    # projections.  FLOPs per step: attention 14 L^2 D per edge, projections 24 L D^2 per node.
    flops_attn, flops_proj = 14 * L * L * D * E, 24 * L * D * D * N
    if lists:      # in-projection + its two gradients over the nodes with any edge, out-projection + its two over the receivers
        flops_proj = L * D * D * (18 * lists['any'] + 6 * lists['in'])
    mfma_peak = MFMA_PEAK_TFLOPS[dt_name]
    planes_ran = any(n.endswith(('_planes', '_scaled')) for n in timer.entry_points)
    # the arithmetic peak of the dominant EDGE kernel: fp32-input MFMA = the vector rate; the plane-format passes issue
    # three 16-bit matrix products per fp32 product
    edge_peak = MFMA_PEAK_TFLOPS['bf16'] / 3 if planes_ran else mfma_peak
    t_step = dt / steps
    t_hbm, t_mfma = b_alg / (HBM_PEAK_GBS * 1e9), (flops_attn + flops_proj) / (mfma_peak * 1e12)
    traffic = traffic_source = None                        # PMC-measured HBM bytes per launch
    try:                                                   # (profiles/pmc_traffic.json, same workload)
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
        if pmc.get('workload') == workload and dt_name == 'f32':
            traffic = pmc['kernels'][dom]['traffic_bytes']
            traffic_source = ('profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload '
                              '(committed profile, gfx950 corrections applied); not measured in this run')
    except (OSError, KeyError, ValueError):
        pass
    edge_ms = sum(ms[k] for k in EDGE_CALLS if k in ms)
    edge_bytes = (6 * E + 9 * N) * R
    out.update({
        'layer_hbm': {'algorithmic_bytes_per_step': b_alg,
                      'achieved_GBps_per_gpu': b_alg * steps / dt / 1e9,
                      'frac_of_8TBps': b_alg * steps / dt / 1e9 / HBM_PEAK_GBS},
        # the three edge kernels together, SURVEY.md 8d's edge-phase figure (6E + 9N) R against their summed time
        'edge_phase_hbm': {'algorithmic_bytes_per_step': edge_bytes, 'kernels_ms_sum': edge_ms,
                           'achieved_GBps': edge_bytes / (edge_ms * 1e-3) / 1e9,
                           'frac_of_8TBps': edge_bytes / (edge_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        'layer_flops': {'attention_per_step': flops_attn, 'projections_per_step': flops_proj,
                        'achieved_TFLOPs_per_gpu': (flops_attn + flops_proj) / t_step / 1e12,
                        'peak_TFLOPs': mfma_peak, 'frac_of_peak': (flops_attn + flops_proj) / t_step / 1e12 / mfma_peak,
                        'peak_note': ('fp32-input MFMA = fp32 vector rate (157.3 TF): the peak of the edge kernels; the '
                                      "'native' projections run as 3 fp16 MFMA products per fp32 product on scaled operands "
                                      '(2.5 PF / 3 = 833 TF fp32-equivalent at the dense 16-bit peak; 6 bf16 products with '
                                      'AMPCONV_PROJ_SCALED=0): node_phase_mfma') if dt_name == 'f32' else 'dense bf16 MFMA',
                        'floor_ms': {'hbm_at_8TBps': 1e3 * t_hbm, 'mfma_at_peak': 1e3 * t_mfma},
                        'binding': 'mfma' if t_mfma > t_hbm else 'hbm'},
        'node_phase_ms': 1e3 * t_step - edge_ms,             # projections + glue + CSR build (+ all-reduce)
        'node_phase_mfma': _node_phase_mfma(flops_proj, t_step - edge_ms * 1e-3, dt_name, args.gemm),
        'kernels_ms': ms, 'per_rank_ms': _per_rank(by_rank, dist_on),
        'allreduce_ms': ar_ms,
        'timing': 'value = wall clock over the K steps between barrier+synchronize fences, max over ranks '
                  '(the contract); kernels_ms = mean HIP-event duration per launch over the same steps (rank 0; every '
                  'rank: per_rank_ms); ms_per_step_hip_events_median = median of per-step HIP-event durations on rank 0 '
                  '(SURVEY 8d protocol)',
        'ms_per_step_hip_events_median': median_ms,
        'roofline': {'kernel': dom, 'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                     'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                     'traffic_source': traffic_source,
                     'algorithmic_bytes_per_launch': alg[dom], 'avg_launch_ms': ms[dom],
                     # the same launch against the arithmetic peak (fp32: MFMA and VALU share the FP32 pipe)
                     'arithmetic': {'flops_per_launch': kflops[dom], 'achieved_TFLOPs': kflops[dom] / (ms[dom] * 1e-3) / 1e12,
                                    'peak_TFLOPs': edge_peak,
                                    'frac': kflops[dom] / (ms[dom] * 1e-3) / 1e12 / edge_peak}},
    })
    if alt_out is not None:
        out['alt_gemm'] = alt_out
    if lists:
        out['node_lists'] = dict(lists, N=N, note='projections run over the listed nodes only (nodes with an in-edge / an '
                                 'out-edge / either); layer_flops and node_phase_mfma count those rows; the HBM figures keep '
                                 "SURVEY 8d's all-node bytes")
    return out


def measure_graphed(workload, steps, warmup, args, dev):
    """The fixed-graph regime (the reference's full-graph Cora runs, experiments/cora_benchmark_full.py): forward and backward
    of the layer recorded once as two HIP graphs (ampnet_amd.GraphedAMPConv) and replayed per step; the graph preparation
    runs once, outside the steps -- unlike measure_full, whose every step builds its CSR / CSC (the GraphSAINT regime).
    Same layer, same kernels, same results bit for bit (tests/test_gpu_parity.py::test_graphed_layer_matches_eager)."""
    from ampnet_amd import GraphedAMPConv, graph_cache
    N, E, L, D, H, desc = WORKLOADS[workload]
    layer, _ = make_layer(D, H, torch.float32, dev, args, False)
    x, ei, dy = make_batch(N, E, L, D, 1234, dev)
    x.requires_grad_(True)
    fast = GraphedAMPConv(layer, x, ei)

    def step():
        layer.zero_grad(set_to_none=True)
        x.grad = None
        fast(x).backward(dy)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del fast, x, dy, ei
    graph_cache.clear()
    return {'metric': 'AMPConv edges/sec (fwd+bwd), fixed graph, forward and backward replayed as HIP graphs',
            'value': E * steps / dt, 'unit': 'edges/s', 'steps': steps, 'ms_per_step': 1e3 * dt / steps,
            'config': {'workload': f'{workload}: {desc}; graph prepared once, outside the steps', 'N': N, 'E': E, 'L': L,
                       'D': D, 'H': H}}


def _projection_note(dt_name, gemm):
    """config.projections: which arithmetic the per-node projections of this line ran."""
    from ampnet_amd.conv import functional as F_
    if dt_name == 'bf16':
        return 'bf16 storage: one bf16 matrix product, fp32 accumulate' + (', node lists on graphs with edge-less nodes'
                                                                            if F_.NODE_LISTS else '')
    if gemm != 'native':
        return {'fp32': 'library fp32 GEMMs (rocBLAS)', 'bf16x3': "hipBLASLt's 3-product bf16 split"}[gemm]
    return ('fp32 in / out / accumulate; operands of >= 2^24 elements: two fp16 planes of the power-of-two-scaled operand, '
            '3 matrix products; smaller: three bf16 planes, 6 products' if F_.PROJ_SCALED
            else 'fp32 in / out / accumulate; three bf16 planes, 6 matrix products')


def _node_phase_mfma(flops_proj, t_node, dt_name, gemm):
    """The node phase against ITS roofline (the matrix pipe).  24 L D^2 per node of fp32-equivalent FLOPs;
    the native fp32 projections issue three fp16 MFMA products per fp32 product (scaled two-plane mode; six bf16 ones
    with AMPCONV_PROJ_SCALED=0), the bf16-storage ones one.  None for the
    library fp32 GEMMs (they run on the fp32 pipe: `layer_flops`)."""
    if t_node <= 0 or not (dt_name == 'bf16' or gemm == 'native'):
        return None
    from ampnet_amd.conv import functional as F_
    issue = (3 if F_.PROJ_SCALED else 6) if dt_name == 'f32' else 1      # (every bench workload that reports this is large)
    peak = MFMA_PEAK_TFLOPS['bf16']
    return {'bound': 'mfma', 'flops_fp32_equivalent': flops_proj, 'matrix_products_per_fp32_product': issue,
            'achieved_TFLOPs_fp32_equivalent': flops_proj / t_node / 1e12,
            'achieved_TFLOPs_16bit_issued': issue * flops_proj / t_node / 1e12, 'peak_TFLOPs': peak,
            'frac': issue * flops_proj / t_node / 1e12 / peak,
            'note': 'lower bound of the rate: node_phase_ms also holds the CSR build and glue; DESIGN.md 4a: power-limited'}


INFINITY_CACHE_BYTES = 256 << 20     # MI355X_MICROARCH.md: die-level L3


def _brief(m):
    """What an extra workload contributes to the headline's line.  A workload whose gathered rows (Q | K | V and dObar of
    every node) fit the 256 MiB Infinity Cache is served on-die after the first touch: algorithmic bytes / time is then not
    a share of the HBM roofline, so no such share is printed for it (`cache_resident`: true)."""
    r = m['roofline']
    c = m['config']
    gathered = c['N'] * c['L'] * c['D'] * 4 * (2 if m['dtype'] == 'bf16' else 4)     # Q|K|V + dObar rows
    out = {'workload': c['workload'], 'dtype': m['dtype'], 'value': m['value'], 'unit': m['unit'],
           'steps': m['steps'], 'ms_per_step': m['ms_per_step'], 'kernels_ms': m['kernels_ms'],
           'edge_entry_points': c.get('edge_entry_points')}
    if gathered <= INFINITY_CACHE_BYTES:
        out['cache_resident'] = True
        out['dominant_kernel'] = {'kernel': r['kernel'], 'avg_launch_ms': r['avg_launch_ms']}
    else:
        out['edge_phase_frac_of_8TBps'] = m['edge_phase_hbm']['frac_of_8TBps']
        out['dominant_kernel'] = {'kernel': r['kernel'], 'frac': r['frac'], 'avg_launch_ms': r['avg_launch_ms']}
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        launch_ranks(args.gpus)                               # never returns

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU '
                         f'(python bench.py --gpus N starts them itself)')
    backend = os.environ.get('AMPCONV_DIST_BACKEND', 'nccl')       # nccl == RCCL on ROCm (gloo: CPU-side tests)
    if os.environ.get('AMPCONV_BENCH_LAUNCH_ONLY') == '1':
        # rehearsal of the launch path on a machine without a GPU (tests/test_bench_launch.py): the
        # ranks rendezvous, all-reduce one CPU tensor over gloo and rank 0 reports; nothing is measured.
        # AMPCONV_BENCH_FAIL_RANK=r: that rank dies right after the rendezvous (the others then sit in a
        # collective that never completes -- what the launcher has to clean up)
        if world > 1:
            dist.init_process_group('gloo')
        if os.environ.get('AMPCONV_BENCH_FAIL_RANK') == str(rank):
            os._exit(3)
        t = torch.ones(1)
        if world > 1:
            dist.all_reduce(t)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({'launch_only': True, 'n_gpus': world, 'ranks_seen': int(t.item())}), flush=True)
        return
    n_dev = torch.cuda.device_count()
    assert n_dev > 0 and torch.cuda.is_available(), 'bench.py needs a GPU'
    if backend == 'nccl' and world > n_dev:
        raise SystemExit(f'bench.py: {world} ranks but {n_dev} GPUs visible (one process per GPU over RCCL)')
    local_rank = int(os.environ.get('LOCAL_RANK', 0)) % n_dev     # gloo rehearsal: ranks may share a card
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # AMPCONV_BENCH_FORCE_DIST=1: join a process group even as the only rank, so that a one-GPU box executes the RCCL
    # code path of the N-rank run (init with device_id, parameter broadcast, gradient all-reduce, barriers)
    dist_on = world > 1 or os.environ.get('AMPCONV_BENCH_FORCE_DIST') == '1'
    if dist_on:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:                   # single forced rank: any free port
            import socket
            with socket.socket() as s:
                s.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(s.getsockname()[1])
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        with _stdout_to_stderr():
            dist.init_process_group(backend, **({'device_id': dev} if backend == 'nccl' else {}))
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)                             # communicator (and its banner) up before anything is timed
            dist.barrier()
            torch.cuda.synchronize()

    dt_name = args.dtype or ('bf16' if args.workload == 'cfg5' else 'f32')
    if args.workload in SAINT:
        out = measure_saint(args.workload, args.steps, args.warmup, args, rank, world, dev, dist_on, dt_name)
    else:
        alt = (world == 1 and dt_name == 'f32' and args.gemm == 'native' and not args.no_alt_gemm
               and not args.softmax_free)
        out = measure_full(args.workload, args.steps, args.warmup, args, rank, world, dev, dist_on, dt_name, alt=alt)
        side = not args.no_extra and not args.softmax_free and args.dtype is None
        saint_of = {'cfg4': 'cfg4-saint', 'tiny': 'tiny-saint'}
        if side and args.workload in saint_of:
            # BASELINE config 4 as written, on the same ranks (every N, one rank included): the mode whose scaling curve
            # can bend (27 ms steps, sampler read-backs, a 1 MB all-reduce)
            s = measure_saint(saint_of[args.workload], max(20, args.steps), max(2, args.warmup), args, rank, world, dev,
                              dist_on, 'f32')
            out['saint'] = {k: s[k] for k in ('value', 'unit', 'ms_per_step', 'steps', 'nodes_avg', 'edges_avg',
                                              'sampler_ms', 'allreduce_ms', 'kernels_ms', 'per_rank_ms')}
            out['saint']['workload'] = s['config']['workload']
        if side and world == 1 and args.workload in ('cfg4', 'tiny'):
            # the other configurations, driver-observed: a few steps each (the small ones: more steps, they are
            # launch-bound and cheap).  `tiny`: the same code path on smoke-sized stand-ins (tests)
            extra = ((('cfg3', 3), ('cfg5', 3), ('cfg3-L1', 20), ('cfg3-L4', 10), ('cora', 50), ('ampgcn-default', 3))
                     if args.workload == 'cfg4' else (('cora', 5),))
            out['extra_workloads'] = {
                w: _brief(measure_full(w, k, 1 if k < 20 else 3, args, rank, world, dev, dist_on,
                                       'bf16' if w == 'cfg5' else 'f32'))
                for w, k in extra}
            # the same Cora-sized layer on a FIXED graph, its launches recorded and replayed (HIP graphs)
            out['extra_workloads']['cora-graphed'] = measure_graphed('cora', 200 if args.workload == 'cfg4' else 5, 5, args, dev)
        if world == 1 and not args.no_cpu_baseline and not args.softmax_free and rank == 0:
            N, E, L, D, H, _ = WORKLOADS[args.workload]
            out['cpu_baseline'] = cpu_baseline(L, D, H, E / N)
    if dist_on:
        # evidence, in the line itself, that the collective library saw every rank: an all-reduce of ones
        seen = torch.ones(1, device=dev)
        dist.all_reduce(seen)
        out['dist'] = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
                       'ranks_seen': int(seen.item())}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
