import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')

# fp32 tolerance of SURVEY.md section 8c: FLAT atol 1e-5, rtol 1e-4 on y, dx, attention weights / outputs;
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


# Per-row tensors (y, dx, attention weights / outputs, messages, logits, embeddings) are held to the FLAT atol of
# SURVEY.md 8c.  Everything else a test compares is a parameter gradient -- a sum over all N*L node-token rows -- whose
# absolute tolerance scales with the magnitude of the gradient.
_FLAT = ('y', 'dx', 'dxq', 'dxkv', 'attn_output', 'attn_output_weights', 'message', 'w', 'logits', 'conv1_embedding',
         'conv2_embedding')


def _is_flat(name):
    """'y', 'dx vs oracle', 'y (mfma vs generic)', 'cfg4: dx[17]' -> per-row tensor; anything else -> parameter gradient."""
    import re
    m = re.match(r'[A-Za-z_0-9]+', name.split(':', 1)[1].strip() if ':' in name else name.strip())
    return bool(m) and m.group(0) in _FLAT


def assert_close_scaled(got, want, name, atol=ATOL, rtol=RTOL, scaled=None):
    """allclose at `atol + rtol * |want|`; for accumulated quantities (parameter gradients: every name that is not one of
    the per-row tensors of _FLAT, or `scaled=True`) atol is multiplied by max(1, max|want|).  Prints the achieved maximum error and the share of the tolerance it
    uses, so the head-room is visible in the test log (pytest -rP)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f'{name}: shape {got.shape} vs {want.shape}'
    if scaled is None:
        scaled = not _is_flat(name)
    scale = max(1.0, float(np.abs(want).max()) if want.size else 1.0) if scaled else 1.0
    err = np.abs(got - want)
    tol = atol * scale + rtol * np.abs(want)
    used = float((err / tol).max()) if err.size else 0.0
    print(f'[tol] {name}: max err {float(err.max()) if err.size else 0.0:.3e}, {100 * used:.1f} % of the tolerance '
          f'({"scaled" if scaled else "flat"} atol {atol * scale:.1e}, max |want| {float(np.abs(want).max()) if want.size else 0.0:.3e})')
    bad = err > tol
    assert not bad.any(), (f'{name}: {int(bad.sum())}/{bad.size} out of tolerance, '
                           f'max err {err.max():.3e} ({"scaled" if scaled else "flat"} atol {atol * scale:.1e})')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN_DIR
