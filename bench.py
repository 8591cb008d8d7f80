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

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg4|cfg3|cfg5|cora|tiny] [--dtype f32|bf16] [--gemm native|fp32|bf16x3]

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel against
the 8 TB/s HBM3E spec peak with durations measured live by HIP events on the
launch stream over the timed region; `cpu_baseline` is the reference-shaped CPU
restatement (oracle/ampconv_torch.py) timed on the host cores on a bounded
sub-sample of the same workload (rank 0, N=1 only).
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
        self.enabled = False

    def wrap(self, lib, name):
        fn = getattr(lib, name)

        def timed(*args):
            if not self.enabled:
                return fn(*args)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            self.events.setdefault(name, []).append((e0, e1))
            return rc
        return timed

    def summary(self):
        return {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in self.events.items() if v}


class TimedLib:
    def __init__(self, lib, timer):
        self._lib = lib
        for n in ('ampconv_fwd_edge', 'ampconv_bwd_edge_dst', 'ampconv_bwd_edge_src'):
            setattr(self, n, timer.wrap(lib, n))

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


def cpu_baseline(L, D, H, ratio):
    """Reference-shaped CPU path (gather -> nn.MultiheadAttention -> scatter-mean), fwd+bwd,
    on a bounded sub-sample with the workload's L, D, H and E/N."""
    from oracle.ampconv_torch import RefShapedAMPConv
    cores = host_cores()
    torch.set_num_threads(cores)
    n_cpu = 3000
    e_cpu = int(n_cpu * ratio)
    torch.manual_seed(1)
    layer = RefShapedAMPConv(D, H)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n_cpu, L * D, generator=g, requires_grad=True)
    ei = torch.randint(0, n_cpu, (2, e_cpu), generator=g)
    dy = torch.randn(n_cpu, L * D, generator=g)

    def step(edges):
        layer.zero_grad(set_to_none=True)
        x.grad = None
        y = layer(x, edges)
        (y * dy).sum().backward()

    step(ei[:, : e_cpu // 8])                 # warm-up on an eighth of the sample
    best = float('inf')
    for _ in range(2):
        t0 = time.perf_counter()
        step(ei)
        best = min(best, time.perf_counter() - t0)
    return {'value': e_cpu / best, 'unit': 'edges/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n_cpu} nodes / {e_cpu} edges, L={L} D={D} H={H} fp32, fwd+bwd, best of 2, '
                      f'torch {torch.__version__} CPU, {cores} threads'}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default=os.environ.get('AMPCONV_BENCH_WORKLOAD', 'cfg4'))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt-gemm', action='store_true',
                    help='skip the extra pass with the projections on the library GEMMs (profiling runs)')
    ap.add_argument('--softmax-free', action='store_true',
                    help="the reference's softmax-free attention variant (AMPConv(..., softmax=False)); not the headline")
    ap.add_argument('--dtype', default=None, choices=['f32', 'bf16'],
                    help='storage dtype (default: f32; bf16 for cfg5)')
    ap.add_argument('--gemm', default='native', choices=['native', 'fp32', 'bf16x3'],
                    help="per-node projections: 'native' (default, the headline: libampconv's own kernels, fp32 operands "
                         "split exactly into three bf16 terms, six partial products on the bf16 matrix cores, fp32-grade "
                         "error -- csrc/proj_gemm.hip), 'fp32' (library GEMMs, rocBLAS) or 'bf16x3' (hipBLASLt's "
                         "3-product split, 8x the error; ampnet_amd.conv.functional.gemm_precision)")
    return ap.parse_args()


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: start N fresh worker processes of this
    script (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) BEFORE this
    process touches the GPU, wait for them, and exit non-zero if any of them failed.  Mirrors the
    reference's own mp.spawn of one process per rank
    (experiments/cora_benchmark_graphsaint_distributed.py:130-142).  Rank 0 prints the JSON line."""
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
    rcs = [p.wait() for p in procs]
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print(f'bench.py: ranks failed (rank, exit code): {bad}', file=sys.stderr)
        sys.exit(1)
    sys.exit(0)


def run_saint(args, rank, world, dev, layer, reducer, N, E, L, D, H, desc, dt_name, tdt, R, dist_on):
    """GraphSAINT-batch mode: per rank per step GraphSAINTRandomWalkSampler.sample() -> row gather of the
    resident features -> AMPConv forward + backward on the induced subgraph -> gradient all-reduce.
    `value` = sampled edges processed by all ranks per second; the sampler's share of the step and the
    host read-backs it costs (sub-graph sizes, needed to size the batch's tensors) are reported."""
    import types
    from ampnet_amd import GraphSAINTRandomWalkSampler, graph_cache, _lib
    roots, walk = SAINT[args.workload]
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
    ev = []
    sizes = []

    def step(timed):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        node_idx, ei_sub, _, _ = sampler.sample()
        e1.record()
        graph_cache.clear()
        layer.zero_grad(set_to_none=True)
        xs = x.index_select(0, node_idx).requires_grad_(True)
        y = layer(xs, ei_sub)
        y.backward(dy_buf[: xs.size(0)])
        if dist_on:
            reducer.allreduce()
        if timed:
            ev.append((e0, e1))
            sizes.append((int(node_idx.numel()), int(ei_sub.size(1))))

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    dt = time.perf_counter() - t0
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
    if rank != 0:
        return
    ms = timer.summary()
    n_avg = sum(n for n, _ in sizes) / len(sizes)
    e_avg = sum(e for _, e in sizes) / len(sizes)
    alg = {'ampconv_fwd_edge': (2 * e_avg + 2 * n_avg) * R, 'ampconv_bwd_edge_dst': (2 * e_avg + 3 * n_avg) * R,
           'ampconv_bwd_edge_src': (2 * e_avg + 4 * n_avg) * R}
    dom = max(ms, key=ms.get)
    achieved = alg[dom] / (ms[dom] * 1e-3) / 1e9
    sampler_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    flops = 14 * L * L * D * e_avg + 24 * L * D * D * n_avg
    t_step = dt / args.steps
    out = {
        'metric': 'AMPConv edges/sec (fwd+bwd), GraphSAINT subgraph batches', 'value': edges_all / dt, 'unit': 'edges/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * t_step,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dt_name, 'data': 'synthetic',
        'config': {'workload': f'{args.workload}: {desc}; sampler + feature-row gather + one AMPConv layer fwd+bwd incl. CSR '
                               f'build of the subgraph' + (' + RCCL grad all-reduce' if world > 1 else ''),
                   'N': N, 'E': E, 'L': L, 'D': D, 'H': H, 'roots': roots, 'walk_length': walk,
                   'parallelism': f'dp{world}', 'gemm': args.gemm if dt_name == 'f32' else 'bf16'},
        'batch': {'nodes_avg': n_avg, 'edges_avg': e_avg, 'sampler_ms': sampler_ms,
                  'sampler_share_of_step': sampler_ms / (1e3 * t_step),
                  'host_readbacks_per_batch': 'sampler: sub-graph node and edge counts (2), CSR build: bounds flag + '
                                              'long-segment chunk counts (1); all inside the timed step'},
        'layer_flops': {'per_step': flops, 'achieved_TFLOPs_per_gpu': flops / t_step / 1e12,
                        'peak_TFLOPs': MFMA_PEAK_TFLOPS[dt_name],
                        'frac_of_peak': flops / t_step / 1e12 / MFMA_PEAK_TFLOPS[dt_name]},
        'kernels_ms': ms,
        'roofline': {'kernel': dom, 'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': None, 'algorithmic_bytes_per_launch': alg[dom],
                     'avg_launch_ms': ms[dom]},
    }
    print(json.dumps(out), flush=True)


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
        # ranks rendezvous, all-reduce one CPU tensor over gloo and rank 0 reports; nothing is measured
        if world > 1:
            dist.init_process_group('gloo')
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
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group(backend, **({'device_id': dev} if backend == 'nccl' else {}))

    from ampnet_amd import AMPConv, graph_cache, _lib
    from ampnet_amd.conv import functional as F_
    from ampnet_amd.distributed import GradientAllReducer, broadcast_parameters

    N, E, L, D, H, desc = WORKLOADS[args.workload]
    dt_name = args.dtype or ('bf16' if args.workload == 'cfg5' else 'f32')
    tdt = torch.bfloat16 if dt_name == 'bf16' else torch.float32
    R = L * D * (2 if dt_name == 'bf16' else 4)
    torch.manual_seed(1)
    layer = AMPConv(D, H, softmax=not args.softmax_free).to(dev)
    layer.retain_attention = False                            # no [E, L, L] side output kept alive
    layer.gemm_precision = args.gemm
    with torch.no_grad():
        layer.multi_head_attention.in_proj_bias.normal_(0, 0.1)
        layer.multi_head_attention.out_proj.bias.normal_(0, 0.1)
    layer = layer.to(tdt)
    if dist_on:
        broadcast_parameters(layer, src=0)
    reducer = GradientAllReducer(layer.parameters())

    if args.workload in SAINT:
        run_saint(args, rank, world, dev, layer, reducer, N, E, L, D, H, desc, dt_name, tdt, R, dist_on)
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        return

    x, ei, dy = make_batch(N, E, L, D, 1234 + rank, dev, tdt, rmat=args.workload == 'cfg5')   # own graph per rank
    x.requires_grad_(True)

    timer = KernelTimer()
    real_lib = _lib.load()
    timed_lib = TimedLib(real_lib, timer)
    orig_load = _lib.load
    _lib.load = lambda: timed_lib                             # functional.py resolves the lib per call

    def step():
        graph_cache.clear()                                   # a new batch -> CSR build is in the step
        layer.zero_grad(set_to_none=True)
        x.grad = None
        y = layer(x, ei)
        y.backward(dy)
        if dist_on:
            reducer.allreduce()

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    timer.enabled = True
    step_events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        step_events.append((e0, e1))
    fence()
    dt = time.perf_counter() - t0
    timer.enabled = False
    _lib.load = orig_load
    step_ms = sorted(a.elapsed_time(b) for a, b in step_events)
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    alt = None
    if world == 1 and dt_name == 'f32' and args.gemm == 'native' and not args.no_alt_gemm and not args.softmax_free:
        # same step with the projections on the library's fp32 GEMMs (rocBLAS) -- what round 2 measured; reported beside
        layer.gemm_precision = 'fp32'
        step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dta = time.perf_counter() - t0
        layer.gemm_precision = args.gemm
        alt = {'gemm': 'fp32 (rocBLAS library GEMMs)', 'value': E * args.steps / dta, 'unit': 'edges/s',
               'ms_per_step': 1e3 * dta / args.steps}

    if rank == 0 and args.softmax_free:
        # the edge phase of this variant is one segment reduction (conv/linear.py): no edge-kernel roofline
        print(json.dumps({'metric': 'AMPConv edges/sec (fwd+bwd), softmax-free variant', 'value': world * E * args.steps / dt,
                          'unit': 'edges/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak',
                          'vs_baseline': None, 'dtype': dt_name, 'data': 'synthetic',
                          'config': {'workload': f'{args.workload}: {desc}, softmax-free AMPConv layer fwd+bwd',
                                     'N': N, 'E': E, 'L': L, 'D': D, 'H': H, 'parallelism': f'dp{world}'}}), flush=True)
    elif rank == 0:
        ms = timer.summary()
        alg = {'ampconv_fwd_edge': (2 * E + 2 * N) * R, 'ampconv_bwd_edge_dst': (2 * E + 3 * N) * R,
               'ampconv_bwd_edge_src': (2 * E + 4 * N) * R}
        kflops = {'ampconv_fwd_edge': 4 * L * L * D * E, 'ampconv_bwd_edge_dst': 6 * L * L * D * E,
                  'ampconv_bwd_edge_src': 8 * L * L * D * E}   # multiply-adds x 2 of the products each pass does
        dom = max(ms, key=ms.get)
        achieved = alg[dom] / (ms[dom] * 1e-3) / 1e9
        value = world * E * args.steps / dt
        b_alg = (6 * E + 15 * N) * R + 16 * E                 # whole layer, SURVEY.md 8d
        # SURVEY.md 8d names two rooflines: HBM for the edge phase, the matrix pipe for the per-node
        # projections.  FLOPs per step: attention 14 L^2 D per edge, projections 24 L D^2 per node.
        flops_attn, flops_proj = 14 * L * L * D * E, 24 * L * D * D * N
        mfma_peak = MFMA_PEAK_TFLOPS[dt_name]
        t_step = dt / args.steps
        t_hbm, t_mfma = b_alg / (HBM_PEAK_GBS * 1e9), (flops_attn + flops_proj) / (mfma_peak * 1e12)
        traffic = None                                         # PMC-measured HBM bytes per launch
        try:                                                   # (profiles/pmc_traffic.json, same workload)
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
            if pmc.get('workload') == args.workload and dt_name == 'f32':
                traffic = pmc['kernels'][dom]['traffic_bytes']
        except (OSError, KeyError, ValueError):
            pass
        out = {
            'metric': 'AMPConv edges/sec (fwd+bwd)', 'value': value, 'unit': 'edges/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': dt_name, 'data': 'synthetic',
            'config': {'workload': f'{args.workload}: {desc}, one AMPConv layer fwd+bwd incl. CSR build, '
                                   f'one graph per GPU' + (' + RCCL grad all-reduce' if world > 1 else ''),
                       'N': N, 'E': E, 'L': L, 'D': D, 'H': H, 'parallelism': f'dp{world}',
                       'gemm': args.gemm if dt_name == 'f32' else 'bf16'},
            'layer_hbm': {'algorithmic_bytes_per_step': b_alg,
                          'achieved_GBps_per_gpu': b_alg * args.steps / dt / 1e9,
                          'frac_of_8TBps': b_alg * args.steps / dt / 1e9 / HBM_PEAK_GBS},
            'layer_flops': {'attention_per_step': flops_attn, 'projections_per_step': flops_proj,
                            'achieved_TFLOPs_per_gpu': (flops_attn + flops_proj) / t_step / 1e12,
                            'peak_TFLOPs': mfma_peak, 'frac_of_peak': (flops_attn + flops_proj) / t_step / 1e12 / mfma_peak,
                            'peak_note': 'fp32-input MFMA = fp32 vector rate (157.3 TF)' if dt_name == 'f32'
                                         else 'dense bf16 MFMA',
                            'floor_ms': {'hbm_at_8TBps': 1e3 * t_hbm, 'mfma_at_peak': 1e3 * t_mfma},
                            'binding': 'mfma' if t_mfma > t_hbm else 'hbm'},
            'kernels_ms': ms,
            'timing': 'value = wall clock over the K steps between barrier+synchronize fences, max over ranks '
                      '(the contract); kernels_ms = mean HIP-event duration per launch over the same steps; '
                      'ms_per_step_hip_events_median = median of per-step HIP-event durations on rank 0 (SURVEY 8d protocol)',
            'ms_per_step_hip_events_median': median_ms,
            'roofline': {'kernel': dom, 'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'algorithmic_bytes_per_launch': alg[dom], 'avg_launch_ms': ms[dom],
                         # the same launch against the arithmetic peak (fp32: MFMA and VALU share the FP32 pipe)
                         'arithmetic': {'flops_per_launch': kflops[dom], 'achieved_TFLOPs': kflops[dom] / (ms[dom] * 1e-3) / 1e12,
                                        'peak_TFLOPs': mfma_peak,
                                        'frac': kflops[dom] / (ms[dom] * 1e-3) / 1e12 / mfma_peak}},
        }
        if alt is not None:
            out['alt_gemm'] = alt
        if world == 1 and not args.no_cpu_baseline:
            del x, dy
            torch.cuda.empty_cache()
            out['cpu_baseline'] = cpu_baseline(L, D, H, E / N)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
