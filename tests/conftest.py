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
# SURVEY.md 8c.  A comparison is magnitude-scaled only when its label NAMES a parameter gradient -- a sum over all N*L
# node-token rows, whose absolute tolerance scales with the magnitude of the gradient (the builder's amendment to 8c, stated
# in DESIGN.md section 2) -- or when the caller says `scaled=True` with its reason at the call.  A new or misspelt label is
# therefore tested at the STRICT flat bar (round-4 advisor finding: it used to fall into the loose class).
_GRADIENT_LABEL = (r'(^|[\s:(])(l\d_)?(g_(in|out)_proj_(weight|bias)|g_(in|out)_[wb]|g[wb]|go[wb]'       # g_in_proj_weight, gw, gob
                   r'|[gd][Wb]_?(in|out|o)?)\b'                                                           # gWin, gb_in, dW_out, db_in
                   r'|parameter gradient|bias gradient|\.grad\b|^grad ')


def _is_flat(name):
    import re
    return re.search(_GRADIENT_LABEL, name) is None


def assert_close_scaled(got, want, name, atol=ATOL, rtol=RTOL, scaled=None):
    """allclose at `atol + rtol * |want|`; for accumulated quantities (parameter gradients: labels that name one, see
    _GRADIENT_LABEL, or `scaled=True`) atol is multiplied by max(1, max|want|).  Prints the achieved maximum error and the share of the tolerance it
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
