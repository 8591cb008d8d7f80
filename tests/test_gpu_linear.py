"""GPU parity of the softmax-free AMPConv variant (ampnet_amd/conv/linear.py):
against the outputs of the reference's own softmax-free attention class (tests/golden/linear_*.npz,
written by oracle/make_golden.py from custom_multihead_attn.py) and against the per-edge numpy
oracle on seeded graphs.  Same fp32 tolerance as the softmax path."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden, assert_close_scaled

pytestmark = pytest.mark.gpu

LINEAR = golden_files(linear=True)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from ampnet_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _layer(g, dev):
    from ampnet_amd import AMPConv
    layer = AMPConv(int(g['D']), int(g['H']), softmax=False).to(dev)
    layer.load_state_dict({'multi_head_attention.in_proj_weight': torch.from_numpy(g['in_proj_weight']),
                           'multi_head_attention.in_proj_bias': torch.from_numpy(g['in_proj_bias']),
                           'multi_head_attention.out_proj.weight': torch.from_numpy(g['out_proj_weight']),
                           'multi_head_attention.out_proj.bias': torch.from_numpy(g['out_proj_bias'])})
    return layer


def _grads(layer):
    m = layer.multi_head_attention
    return [t.grad.cpu().numpy() for t in (m.in_proj_weight, m.in_proj_bias, m.out_proj.weight, m.out_proj.bias)]


@pytest.mark.parametrize('path', LINEAR, ids=[os.path.basename(p)[:-4] for p in LINEAR])
def test_golden_softmax_free(path, dev):
    g = load_golden(path)
    layer = _layer(g, dev)
    x = torch.from_numpy(g['x']).to(dev).requires_grad_(True)
    ei = torch.from_numpy(g['edge_index']).to(dev)
    y = layer(x, ei)
    (y * torch.from_numpy(g['dy']).to(dev)).sum().backward()
    yh = y.detach().cpu().numpy()
    assert_close_scaled(yh, g['y'], 'y')
    deg = np.bincount(g['edge_index'][1], minlength=int(g['N']))
    assert (yh[deg == 0] == 0).all()
    assert_close_scaled(x.grad.cpu().numpy(), g['dx'], 'dx')
    for got, name in zip(_grads(layer), ('g_in_proj_weight', 'g_in_proj_bias', 'g_out_proj_weight', 'g_out_proj_bias')):
        assert_close_scaled(got, g[name], name)
    w = layer.attn_output_weights.cpu().numpy()
    assert_close_scaled(w[g['w_edges']], g['attn_output_weights'], 'attn_output_weights')
    ao = layer.attn_output.cpu().numpy()
    assert_close_scaled(ao[g['w_edges'][:4]], g['attn_output'], 'attn_output')


@pytest.mark.parametrize('shape', [(1500, 14000, 20, 128, 4), (900, 8000, 20, 256, 8), (300, 2500, 40, 100, 2)],
                         ids=['D128', 'D256', 'L40_dh50'])
def test_seeded_vs_oracle_softmax_free(shape, dev):
    from ampnet_amd import AMPConv
    from oracle.ampconv_numpy import AMPConvOracle
    N, E, L, D, H = shape
    g = torch.Generator().manual_seed(N + E)
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[1, : E // 8] = 3                              # a hub destination
    ei[0, E // 8: E // 4] = 5                        # a hub source
    ei[1, ei[1] == 7] = 8                            # node 7 receives nothing
    x = torch.randn(N, L * D, generator=g) * 0.5
    dy = torch.randn(N, L * D, generator=g)
    torch.manual_seed(5)
    layer = AMPConv(D, H, softmax=False).to(dev)
    with torch.no_grad():
        layer.multi_head_attention.in_proj_bias.normal_(0, 0.1)
        layer.multi_head_attention.out_proj.bias.normal_(0, 0.1)
    xg = x.to(dev).requires_grad_(True)
    y = layer(xg, ei.to(dev))
    y.backward(dy.to(dev))
    m = layer.multi_head_attention
    o = AMPConvOracle(m.in_proj_weight.detach().cpu().numpy(), m.in_proj_bias.detach().cpu().numpy(),
                      m.out_proj.weight.detach().cpu().numpy(), m.out_proj.bias.detach().cpu().numpy(), H,
                      softmax=False)
    y_ref, _ = o.forward(x.numpy(), ei.numpy(), need_weights=False)
    ref = o.backward(dy.numpy())
    yh = y.detach().cpu().numpy()
    assert_close_scaled(yh, y_ref, 'y')
    assert (yh[7] == 0).all()
    assert_close_scaled(xg.grad.cpu().numpy(), ref[0], 'dx')
    for got, want, name in zip(_grads(layer), ref[1:], ('gW_in', 'gb_in', 'gW_out', 'gb_out')):
        assert_close_scaled(got, want, name)


def test_message_softmax_free(dev):
    # message(x_i, x_j) on pre-gathered pairs = per-edge softmax-free attention + out-projection
    from ampnet_amd import AMPConv
    from oracle.ampconv_numpy import AMPConvOracle
    E, L, D, H = 60, 6, 32, 4
    g = torch.Generator().manual_seed(3)
    x_i, x_j = torch.randn(E, L * D, generator=g), torch.randn(E, L * D, generator=g)
    torch.manual_seed(9)
    layer = AMPConv(D, H, softmax=False).to(dev)
    out = layer.message(x_i.to(dev), x_j.to(dev)).detach().cpu().numpy()
    m = layer.multi_head_attention
    o = AMPConvOracle(m.in_proj_weight.detach().cpu().numpy(), m.in_proj_bias.detach().cpu().numpy(),
                      m.out_proj.weight.detach().cpu().numpy(), m.out_proj.bias.detach().cpu().numpy(), H,
                      softmax=False)
    # graph with 2E nodes: edge e goes from node E + e (x_j) to node e (x_i)
    xx = np.concatenate([x_i.numpy(), x_j.numpy()])
    ei = np.stack([np.arange(E) + E, np.arange(E)])
    y_ref, _ = o.forward(xx, ei, need_weights=False)
    assert_close_scaled(out, y_ref[:E], 'message')


def test_gather_segment_sum_rejects_bad_arguments(dev):
    from ampnet_amd import _lib
    lib = _lib.load()
    t = torch.zeros(8, device=dev)
    p = torch.zeros(2, dtype=torch.int32, device=dev)
    assert lib.ampconv_gather_segment_sum(t.data_ptr(), p.data_ptr(), p.data_ptr(), None, 0, 1, 6,
                                          t.data_ptr(), None) == -1     # AMPCONV_E_BADARG: F % 4 != 0
    assert lib.ampconv_gather_segment_sum(None, p.data_ptr(), p.data_ptr(), None, 0, 1, 8,
                                          t.data_ptr(), None) == -1
