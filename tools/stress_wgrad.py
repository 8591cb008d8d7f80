"""Developer stress test: the masked / plain weight-gradient product launched 400 times on the same inputs must give the same
bits every time (tests/test_gpu_proj.py::test_proj_wgrad_masked_is_reproducible is the short form)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from ampnet_amd.conv import functional as F_
dev = torch.device('cuda:0')
torch.manual_seed(0)
scaled = '--scaled' in sys.argv
for (M, Na, Nb, L) in ((60000, 128, 128, 20), (60000, 384, 128, 20), (60000, 256, 256, 20), (60000, 768, 256, 20), (40000, 100, 100, 20)):
    a = torch.randn(M, Na, device=dev)
    b = torch.randn(M, Nb, device=dev)
    amax = (F_.absmax(a), F_.absmax(b)) if scaled else None
    deg = (torch.rand(M // L, device=dev) < 0.8).int()
    rp = torch.zeros(M // L + 1, dtype=torch.int32, device=dev)
    rp[1:] = torch.cumsum(deg, 0)
    for masked in (True, False):
        dw0 = torch.empty(Na, Nb, device=dev); cs0 = torch.empty(Na, device=dev)
        F_.proj_wgrad(a, b, dw0, cs0, rp if masked else None, L, amax=amax)
        bad = 0
        for it in range(400):
            dw = torch.empty(Na, Nb, device=dev); cs = torch.empty(Na, device=dev)
            if it % 2:
                junk = torch.randn(1 << 20, device=dev) * 2      # something else on the stream in between
            F_.proj_wgrad(a, b, dw, cs, rp if masked else None, L, amax=amax)
            if not (torch.equal(dw, dw0) and torch.equal(cs, cs0)):
                bad += 1
                if bad <= 3:
                    d = (dw != dw0)
                    print('   mismatch at it', it, int(d.sum()), 'entries; rows', d.any(1).nonzero().flatten()[:4].tolist(), 'cols', d.any(0).nonzero().flatten()[:4].tolist(), flush=True)
        print(M, Na, Nb, 'scaled' if scaled else 'six', 'masked' if masked else 'plain', 'mismatching runs:', bad, '/ 400', flush=True)
