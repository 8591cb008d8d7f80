"""Diagnostic: per-phase cycle shares of proj_rows_kernel from a build with -DAMPCONV_PROJ_STAMPS
(__graft_entry__.build_variant('PSTAMP', ['-DAMPCONV_PROJ_STAMPS'])).  Shares only: a stamped build is slower
than the product (its fences forbid overlaps the real kernel has)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['AMPCONV_LIB_PATH'] = os.path.join(ROOT, 'build', 'variants', 'libampconv_PSTAMP.so')
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ampnet_amd.conv import functional as F_  # noqa: E402

M, D = 4_000_000, 256
dev = torch.device('cuda:0')
x = torch.randn(M, D, device=dev)
w_in = torch.randn(3 * D, D, device=dev) * 0.06
b_in = torch.randn(3 * D, device=dev)
img = F_.proj_image(w_in)
scaled = '--scaled' in sys.argv                 # the two-plane kernels (three products per step instead of six)
amax = F_.absmax(x) if scaled else None
for _ in range(3):
    F_.proj_rows(x, img, b_in, amax=amax)
torch.cuda.synchronize()
raw = ctypes.CDLL(os.environ['AMPCONV_LIB_PATH'])
buf = (ctypes.c_ulonglong * (8 * 4096))()
assert raw.ampconv_debug_read_proj_stamps(buf, 8 * 4096) == 0
a = np.array(buf, dtype=np.float64).reshape(4096, 8)
a = a[a.sum(1) > 0]
names = ['tile start / loop overhead', 'barrier before the even step', 'even step: DMA issue, 48 x NTW/2 MFMAs, split, row request, DMA wait',
         'barrier before the odd step', 'odd step: DMA issue, MFMAs, split, DMA wait', '-', 'store tail (barrier, LDS staging, stores)', '-']
tot = a.sum()
variant = int(os.environ.get('AMPCONV_PROJ_ROWS', '0'))
bm, bn, waves = {0: (128, 256, 4), 1: (256, 256, 8), 2: (128, 128, 4)}[variant]
blocks = (M / bm) * (3 * D / bn) * (D / 32) / len(a)
mfma = 2 * (bm // 32) * (bn // 32) * (3 if scaled else 6) // waves
print(f'shape {variant} ({bm} x {bn}, {waves} waves): {len(a)} workgroups, {blocks:.0f} 32-deep line blocks each, '
      f'{a.sum(1).mean() / blocks:.0f} cycles per block per wave (MFMA issue alone: {mfma * 32}, x2 with the SIMD partner)')
for n, v in zip(names, a.sum(0)):
    print(f'{n:100s} {v / len(a) / blocks:8.0f} cycles/block {100 * v / tot:5.1f} %')
