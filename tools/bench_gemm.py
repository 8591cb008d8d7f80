"""Dev: activation GEMMs of the layer at cfg4 size under rocBLAS vs hipBLASLt."""
import torch
dev = torch.device('cuda:0')
M, D = 20_000_000, 256
x = torch.randn(M, D, device=dev)
w3 = torch.randn(3 * D, D, device=dev) * 0.05
b3 = torch.randn(3 * D, device=dev)
w = torch.randn(D, D, device=dev) * 0.05
g3 = torch.randn(M, 3 * D, device=dev)
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for lib in ('default', 'hipblaslt', 'hipblas'):
    try:
        if lib != 'default':
            torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib, 'unavailable', e); continue
    a = t(lambda: torch.addmm(b3, x, w3.t()))
    b = t(lambda: x.mm(w.t()))
    c = t(lambda: g3.mm(w3))
    print(f'{lib:10s} qkv {a:7.2f} ms ({2*M*D*3*D/a/1e9:5.1f} TF)  out/dobar {b:7.2f} ms ({2*M*D*D/b/1e9:5.1f} TF)  dx {c:7.2f} ms ({2*M*D*3*D/c/1e9:5.1f} TF)')
