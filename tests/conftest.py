import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')

# fp32 tolerance of SURVEY.md section 8c: atol 1e-5, rtol 1e-4 on y, dx, weights;
# parameter gradients are sums over N*L rows, so their atol scales with the
# magnitude of the gradient (see assert_close_scaled).
ATOL, RTOL = 1e-5, 1e-4


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden_files(two_layer=False, linear=False):
    """Fixtures written by oracle/make_golden.py.  linear=True: the softmax-free variant
    (linear_*.npz, from the reference's custom_multihead_attn.py)."""
    fs = sorted(glob.glob(os.path.join(GOLDEN_DIR, '*.npz')))
    fs = [f for f in fs if not os.path.basename(f).startswith(('model_', 'sampler_'))]   # model_files(), sampler fixtures
    fs = [f for f in fs if os.path.basename(f).startswith('linear_') == linear]
    return [f for f in fs if ('2layer' in os.path.basename(f)) == two_layer]


def model_files():
    """Whole-model fixtures written by oracle/make_golden_ampgcn.py from the reference's AMPGCN."""
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, 'model_*.npz')))


def load_golden(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def assert_close_scaled(got, want, name, atol=ATOL, rtol=RTOL):
    """allclose with atol scaled by max|want| for accumulated quantities."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f'{name}: shape {got.shape} vs {want.shape}'
    scale = max(1.0, float(np.abs(want).max()) if want.size else 1.0)
    err = np.abs(got - want)
    tol = atol * scale + rtol * np.abs(want)
    bad = err > tol
    assert not bad.any(), (f'{name}: {int(bad.sum())}/{bad.size} out of tolerance, '
                           f'max err {err.max():.3e} (scale {scale:.3e})')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN_DIR
