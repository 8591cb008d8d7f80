"""Pin both CPU restatements in oracle/ against the reference's own outputs
(tests/golden/*.npz, written by oracle/make_golden.py from the imported
reference) and against the reference's known-answer check
(synthetic_benchmark/testing_message_passing_pyg.py:37-40)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden, assert_close_scaled
from oracle.ampconv_numpy import AMPConvOracle, segment_mean
from oracle.ampconv_torch import RefShapedAMPConv

SINGLE = golden_files(two_layer=False)
TWO = golden_files(two_layer=True)
LINEAR = golden_files(linear=True)


def test_golden_present():
    assert len(SINGLE) >= 10 and len(TWO) >= 1


def test_pyg_mean_known_answer():
    # testing_message_passing_pyg.py:23-40: [6,6,6] without the self loop,
    # [5.4,5.4,5.4] with it, zeros for every node nobody sends to.
    x = np.array([[1, 1, 1], [2, 2, 2], [3, 3, 3], [10, 10, 10], [11, 11, 11]], dtype=np.float32)
    for ei, want in (([[0, 1, 3, 4], [2, 2, 2, 2]], 6.0), ([[0, 1, 2, 3, 4], [2, 2, 2, 2, 2]], 5.4)):
        ei = np.array(ei)
        out = segment_mean(x[ei[0]], ei[1], 5)
        np.testing.assert_allclose(out[2], [want] * 3, rtol=1e-6)
        assert (out[[0, 1, 3, 4]] == 0).all()


@pytest.mark.parametrize('path', SINGLE, ids=[os.path.basename(p)[:-4] for p in SINGLE])
@pytest.mark.parametrize('dtype', [np.float64, np.float32], ids=['f64', 'f32'])
def test_numpy_oracle_vs_reference(path, dtype):
    g = load_golden(path)
    o = AMPConvOracle(g['in_proj_weight'], g['in_proj_bias'], g['out_proj_weight'],
                      g['out_proj_bias'], int(g['H']), dtype=dtype)
    y, w = o.forward(g['x'], g['edge_index'])
    assert_close_scaled(y, g['y'], 'y')
    assert_close_scaled(w[g['w_edges']], g['attn_output_weights'], 'attn_output_weights')
    np.testing.assert_allclose(w.sum(-1), 1.0, atol=1e-5)            # amp_conv.py:43-47
    deg = np.bincount(g['edge_index'][1], minlength=int(g['N']))
    assert (y[deg == 0] == 0).all()                                   # exact zeros
    dx, dWin, dbin, dWo, dbo = o.backward(g['dy'])
    assert_close_scaled(dx, g['dx'], 'dx')
    assert_close_scaled(dWin, g['g_in_proj_weight'], 'g_in_proj_weight')
    assert_close_scaled(dbin, g['g_in_proj_bias'], 'g_in_proj_bias')
    assert_close_scaled(dWo, g['g_out_proj_weight'], 'g_out_proj_weight')
    assert_close_scaled(dbo, g['g_out_proj_bias'], 'g_out_proj_bias')
    ao = o.attn_output()
    assert_close_scaled(ao[g['w_edges'][:4]], g['attn_output'], 'attn_output')


@pytest.mark.parametrize('path', LINEAR, ids=[os.path.basename(p)[:-4] for p in LINEAR])
def test_numpy_oracle_softmax_free_vs_reference(path):
    # the variant amp_conv.py:6,17 refers to: the reference's nn.MultiheadAttention copy with the
    # softmax line removed (custom_multihead_attn_forward.py:4179-4180)
    g = load_golden(path)
    o = AMPConvOracle(g['in_proj_weight'], g['in_proj_bias'], g['out_proj_weight'],
                      g['out_proj_bias'], int(g['H']), softmax=False)
    y, w = o.forward(g['x'], g['edge_index'])
    assert_close_scaled(y, g['y'], 'y')
    assert_close_scaled(w[g['w_edges']], g['attn_output_weights'], 'attn_output_weights')
    deg = np.bincount(g['edge_index'][1], minlength=int(g['N']))
    assert (y[deg == 0] == 0).all()
    dx, dWin, dbin, dWo, dbo = o.backward(g['dy'])
    assert_close_scaled(dx, g['dx'], 'dx')
    assert_close_scaled(dWin, g['g_in_proj_weight'], 'g_in_proj_weight')
    assert_close_scaled(dbin, g['g_in_proj_bias'], 'g_in_proj_bias')
    assert_close_scaled(dWo, g['g_out_proj_weight'], 'g_out_proj_weight')
    assert_close_scaled(dbo, g['g_out_proj_bias'], 'g_out_proj_bias')
    assert_close_scaled(o.attn_output()[g['w_edges'][:4]], g['attn_output'], 'attn_output')


def _torch_layer(g, prefix=''):
    layer = RefShapedAMPConv(int(g['D']), int(g['H']))
    m = layer.multi_head_attention
    with torch.no_grad():
        m.in_proj_weight.copy_(torch.from_numpy(g[prefix + 'in_proj_weight']))
        m.in_proj_bias.copy_(torch.from_numpy(g[prefix + 'in_proj_bias']))
        m.out_proj.weight.copy_(torch.from_numpy(g[prefix + 'out_proj_weight']))
        m.out_proj.bias.copy_(torch.from_numpy(g[prefix + 'out_proj_bias']))
    return layer


@pytest.mark.parametrize('path', SINGLE, ids=[os.path.basename(p)[:-4] for p in SINGLE])
def test_torch_restatement_vs_reference(path):
    g = load_golden(path)
    layer = _torch_layer(g)
    x = torch.from_numpy(g['x']).requires_grad_(True)
    y = layer(x, torch.from_numpy(g['edge_index']))
    (y * torch.from_numpy(g['dy'])).sum().backward()
    assert_close_scaled(y.detach().numpy(), g['y'], 'y')
    assert_close_scaled(x.grad.numpy(), g['dx'], 'dx')
    m = layer.multi_head_attention
    assert_close_scaled(m.in_proj_weight.grad.numpy(), g['g_in_proj_weight'], 'g_in_proj_weight')
    assert_close_scaled(m.out_proj.bias.grad.numpy(), g['g_out_proj_bias'], 'g_out_proj_bias')
    assert_close_scaled(layer.attn_output_weights.detach().numpy()[g['w_edges']],
                        g['attn_output_weights'], 'attn_output_weights')


@pytest.mark.parametrize('path', TWO, ids=[os.path.basename(p)[:-4] for p in TWO])
def test_numpy_oracle_two_layer(path):
    # conv -> ReLU -> conv -> ReLU (src/ampnet/module/amp_gcn.py:248-262)
    g = load_golden(path)
    H = int(g['H'])
    o1 = AMPConvOracle(g['l1_in_proj_weight'], g['l1_in_proj_bias'], g['l1_out_proj_weight'],
                       g['l1_out_proj_bias'], H)
    o2 = AMPConvOracle(g['l2_in_proj_weight'], g['l2_in_proj_bias'], g['l2_out_proj_weight'],
                       g['l2_out_proj_bias'], H)
    h1, _ = o1.forward(g['x'], g['edge_index'], need_weights=False)
    a1 = np.maximum(h1, 0)
    h2, _ = o2.forward(a1, g['edge_index'], need_weights=False)
    y = np.maximum(h2, 0)
    assert_close_scaled(y, g['y'], 'y')
    d2 = g['dy'] * (h2 > 0)
    da1, dW2, db2, dWo2, dbo2 = o2.backward(d2)
    dx, dW1, db1, dWo1, dbo1 = o1.backward(da1 * (h1 > 0))
    assert_close_scaled(dx, g['dx'], 'dx')
    assert_close_scaled(dW1, g['l1_g_in_proj_weight'], 'l1_g_in_proj_weight')
    assert_close_scaled(dW2, g['l2_g_in_proj_weight'], 'l2_g_in_proj_weight')
    assert_close_scaled(dbo1, g['l1_g_out_proj_bias'], 'l1_g_out_proj_bias')
    assert_close_scaled(dWo2, g['l2_g_out_proj_weight'], 'l2_g_out_proj_weight')


# ---- whole-model fixtures generated FROM the reference's AMPGCN (oracle/make_golden_ampgcn.py) pin the
# numpy restatement of the model (featuriser branches, poolings) on the CPU
def _model_cfg(g):
    return dict(zip(g['cfg_keys'].tolist(), g['cfg_vals'].tolist()))


@pytest.mark.parametrize('path', __import__('conftest').model_files(),
                         ids=[os.path.basename(p)[:-4] for p in __import__('conftest').model_files()])
def test_ampgcn_oracle_matches_reference_model(path):
    from conftest import load_golden, assert_close_scaled
    from oracle.ampgcn_numpy import AMPGCNOracle
    g = load_golden(path)
    cfg = _model_cfg(g)
    state = {k[len('param.'):]: v for k, v in g.items() if k.startswith('param.')}
    reps = cfg['feature_repeats']
    o = AMPGCNOracle(state, int(cfg['num_heads']), average_pooling=cfg['average_pooling_flag'] == 'True',
                     feature_repeats=1 if reps == 'None' else int(reps))
    logits = o.forward(g['x'], g['edge_index'], g.get('sampled_node_feat_indices'))
    assert_close_scaled(logits, g['logits'], 'logits')
    assert_close_scaled(o.e1, g['conv1_embedding'], 'conv1_embedding')
    assert_close_scaled(o.e2, g['conv2_embedding'], 'conv2_embedding')
    grads = o.backward(g['dlogits'])
    for k, v in grads.items():
        assert_close_scaled(v, g['grad.' + k], 'grad ' + k)
    if cfg['average_pooling_flag'] == 'False':
        assert 'param.cls_token' in g and 'grad.cls_token' not in g      # defined, never used (amp_gcn.py:55-57)


# ---- sampler fixture generated by the reference's vendored GraphSAINT classes (oracle/make_golden_sampler.py):
# pins the numpy restatement of the deterministic parts (induced subgraph, attribute subsetting, norms)
def test_graphsaint_restatement_matches_reference_sampler():
    from conftest import GOLDEN_DIR, load_golden
    from oracle import graphsaint_numpy as gs
    g = load_golden(os.path.join(GOLDEN_DIR, 'sampler_rw.npz'))
    ei, N = g['edge_index'], int(g['N'])
    E = ei.shape[1]
    # norms: replay the samples __compute_norm__ drew
    node_count, edge_count = np.zeros(N, np.float32), np.zeros(E, np.float32)
    for w in g['norm_walks']:
        assert gs.walk_is_valid(ei, N, w)
        node_idx, _, keep = gs.induced_subgraph(ei, N, w)
        node_count[node_idx] += 1
        edge_count[keep] += 1
    assert node_count.sum() >= N * int(g['sample_coverage'])                 # the reference's stopping rule
    num_samples = len(g['norm_walks'])
    node_norm, edge_norm = gs.norms(node_count, edge_count, ei[0], N, num_samples)
    np.testing.assert_allclose(node_norm, g['node_norm'], rtol=1e-6)
    np.testing.assert_allclose(edge_norm, g['edge_norm'], rtol=1e-6)
    # batches: same node sets, same induced edges (as multisets of (row, col, edge id))
    x = np.arange(N, dtype=np.float32).reshape(N, 1) * 2.0
    attr = np.arange(E, dtype=np.float32) + 0.5
    for i, w in enumerate(g['epoch_walks']):
        node_idx, e_sub, keep = gs.induced_subgraph(ei, N, w)
        assert node_idx.size == int(g[f'b{i}_num_nodes'])
        np.testing.assert_array_equal(x[node_idx], g[f'b{i}_x'])
        np.testing.assert_allclose(node_norm[node_idx], g[f'b{i}_node_norm'], rtol=1e-6)
        want = sorted(zip(g[f'b{i}_edge_index'][0].tolist(), g[f'b{i}_edge_index'][1].tolist(), g[f'b{i}_edge_attr'].tolist()))
        got = sorted(zip(e_sub[0].tolist(), e_sub[1].tolist(), attr[keep].tolist()))
        assert got == want
