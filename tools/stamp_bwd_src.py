"""Diagnostic: per-phase cycle shares of bwd_src_mfma from a build with -DAMPCONV_STAMPS
(__graft_entry__.build_variant('STAMP', ['-DAMPCONV_STAMPS'])).  Shares only -- a stamped build is slower than the product."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['AMPCONV_LIB_PATH'] = os.path.join(ROOT, 'build', 'variants', 'libampconv_STAMP.so')
import numpy as np, torch
from ampnet_amd import _lib, EdgeCSR
from ampnet_amd.conv import functional as F_
N, E, L, D, H = 100000, 1000000, 20, 256, 8
dev = torch.device('cuda:0'); lib = _lib.load(); dh = D // H
qkv = torch.randn(N * L, 3 * D, device=dev); dobar = torch.randn(N * L, D, device=dev)
ei = torch.randint(0, N, (2, E), device=dev); csr = EdgeCSR(ei, N)
Qv, Kv, Vv = (F_._view(qkv, i * D, L, dh) for i in range(3))
dqkv = torch.empty(N * L, 3 * D, device=dev); dQv, dKv, dVv = (F_._view(dqkv, i * D, L, dh) for i in range(3))
dOv = F_._view(dobar, 0, L, dh); st = torch.cuda.current_stream().cuda_stream
T4 = '--t4' in sys.argv        # the shipped kernel (bwd_src_mfma_t4, statistics from the destination pass) instead of the round-1 one
stats = None
if T4:
    stats = torch.empty(lib.ampconv_softmax_stats_bytes(E, L, D, H, 0) // 4, device=dev)
    spos = csr.csc_positions()
    _lib.check(lib.ampconv_bwd_edge_dst(Qv, Kv, Vv, dOv, csr.rowptr.data_ptr(), csr.col.data_ptr(), N, L, D, H, dQv,
                                        None, 0, None, spos.data_ptr(), stats.data_ptr(), None, 0, st), 'dst')
for _ in range(2):
    _lib.check(lib.ampconv_bwd_edge_src(Qv, Kv, Vv, dOv, csr.cscptr.data_ptr(), csr.crow.data_ptr(), csr.cinv.data_ptr(),
                                        N, L, D, H, dKv, dVv, None, 0, None, stats.data_ptr() if T4 else None, None, 0 if T4 else 2, st), 'src')
torch.cuda.synchronize()
raw = ctypes.CDLL(os.environ['AMPCONV_LIB_PATH'])
buf = (ctypes.c_ulonglong * (8 * 4096))()
assert raw.ampconv_debug_read_stamps(buf, 8 * 4096) == 0
a = np.array(buf, dtype=np.float64).reshape(4096, 8)
deg = (csr.cscptr[1:] - csr.cscptr[:-1]).cpu().numpy()[:512].repeat(8)
if T4:
    per_edge = a.sum(0) / deg.sum()
    names = ['top of edge (loop, previous barrier)', 'tiles -> LDS (incl. vmcnt wait)', 'fetch issue + barrier',
             'S: lds reads, 8+8 MFMA, exp', 'dP: lds reads, 8+8 MFMA, dS', 'dV,dK: 16 lds reads + 16 MFMA + 4x4x1 phase',
             'tail tile of the batch (per edge: /4)', 'unit epilogue start']
    tot = per_edge.sum()
    for n, v in zip(names, per_edge):
        print(f'{n:44s} {v:9.0f} cycles/edge  {100 * v / tot:5.1f} %')
    print('total per edge per wave', tot)
    sys.exit(0)
per_edge = a[:, :6].sum(0) / deg.sum()
names = ['loop overhead/prev tail', 'ds_write (incl. vmcnt wait)', 'fetch issue (scalar+vmem)', 'S,dP: lds reads + 64 MFMA',
         'softmax (DPP)', 'dV,dK: lds reads + 40 MFMA']
tot = per_edge.sum()
for n, v in zip(names, per_edge):
    print(f'{n:32s} {v:9.0f} cycles/edge  {100 * v / tot:5.1f} %')
print('total per edge per wave', tot)
