"""ctypes binding of libampconv.so (C ABI declared in include/ampconv.h).

The product path has no fallback: if the shared library is missing or does not
export a symbol, importing/using the HIP path raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('AMPCONV_LIB_PATH', os.path.join(_HERE, 'libampconv.so'))   # override: dev A/B builds

EXPECTED_ABI = 107          # AMPCONV_VERSION of include/ampconv.h this binding was written against

AMPCONV_F32 = 0
AMPCONV_BF16 = 1
HUB_CHUNK = int(os.environ.get('AMPCONV_HUB_CHUNK', 0))    # edges per chunk of a long CSR/CSC segment (include/ampconv.h, long segments); 0 = by size


def hub_chunk(num_edges):
    """Edges per chunk of a long segment.  64 on small graphs (a chunk is walked by ONE wave, so it bounds the launch's
    latency: ~1.5 us per edge), 128 from a million edges up (cfg5: 619 -> 611 ms per step, half the partial-tile
    workspace; 128-256 is a flat optimum once the main pass is XCD-balanced, tools/sweep_hub_chunk.sh)."""
    return HUB_CHUNK if HUB_CHUNK > 0 else (128 if num_edges >= (1 << 20) else 64)
COLSUM_BLOCKS = 1024      # scratch blocks of ampconv_masked_colsum (csrc/node_ops.hip)


class View(ctypes.Structure):
    """ampconv_view_t: element (n, l, h, c) at ptr + n*node + l*row + h*head + c."""
    _fields_ = [('ptr', ctypes.c_void_p), ('node_stride', ctypes.c_int64),
                ('row_stride', ctypes.c_int64), ('head_stride', ctypes.c_int64)]


_vp, _i64, _i32, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t


class WeightImage(ctypes.Structure):
    """ampconv_weight_image_t: one job of ampconv_proj_weight_images."""
    _fields_ = [('W', ctypes.c_void_p), ('stride_n', ctypes.c_int64), ('stride_k', ctypes.c_int64),
                ('N', ctypes.c_int), ('K', ctypes.c_int), ('image', ctypes.c_void_p)]

# name -> (restype, argtypes); mirrors include/ampconv.h one to one
SIGNATURES = {
    'ampconv_version': (_i32, []),
    'ampconv_error_string': (ctypes.c_char_p, [_i32]),
    'ampconv_csr_workspace_bytes': (_sz, [_i64, _i64]),
    'ampconv_csr_build': (_i32, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'ampconv_graph_build': (_i32, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    'ampconv_csc_positions_from': (_i32, [_vp, _vp, _i64, _vp, _vp]),
    'ampconv_hub_plan_bytes': (_sz, [_i64, _i32]),
    'ampconv_hub_plan': (_i32, [_vp, _i64, _i64, _i32, _vp, _vp]),
    'ampconv_hub_workspace_bytes': (_sz, [_i64, _i32, _i32, _i32]),
    'ampconv_fwd_edge': (_i32, [View, View, View, _vp, _vp, _vp, _i64, _i32, _i32, _i32, View, _vp, _i64, _vp, _i32, _vp]),
    'ampconv_csc_positions': (_i32, [_vp, _vp, _i64, _vp, _vp, _vp]),
    'ampconv_softmax_stats_bytes': (_sz, [_i64, _i32, _i32, _i32, _i32]),
    'ampconv_bwd_edge_dst': (_i32, [View, View, View, View, _vp, _vp, _i64, _i32, _i32, _i32, View, _vp, _i64, _vp,
                                    _vp, _vp, _vp, _i32, _vp]),
    'ampconv_bwd_edge_src': (_i32, [View, View, View, View, _vp, _vp, _vp, _i64, _i32, _i32, _i32,
                                    View, View, _vp, _i64, _vp, _vp, _vp, _i32, _vp]),
    'ampconv_attn_weights': (_i32, [View, View, _vp, _i64, _i32, _i32, _i32, _vp, _i32, _vp]),
    'ampconv_attn_scores': (_i32, [View, View, _vp, _i64, _i32, _i32, _i32, _vp, _i32, _vp]),
    'ampconv_gather_segment_sum': (_i32, [_vp, _vp, _vp, _vp, _i32, _i64, _i64, _vp, _vp]),
    'ampconv_linear_outer': (_i32, [View, View, _i64, _i32, _i32, _i32, ctypes.c_float, _vp, _vp]),
    'ampconv_linear_apply': (_i32, [View, _vp, _i32, _i64, _i32, _i32, _i32, ctypes.c_float, View, _vp]),
    'ampconv_saint_random_walk': (_i32, [_vp, _vp, _vp, _i64, _i32, ctypes.c_uint64, _vp, _vp]),
    'ampconv_saint_workspace_bytes': (_sz, [_i64]),
    'ampconv_saint_nodes': (_i32, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'ampconv_saint_count_edges': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'ampconv_saint_count_edges_bounded': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'ampconv_saint_fill_edges': (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    'ampconv_saint_add_counts': (_i32, [_vp, _i64, _vp, _vp]),
    'ampconv_saint_norms': (_i32, [_vp, _vp, _vp, _i64, _i64, ctypes.c_float, _vp, _vp, _vp]),
    'ampconv_saint_gather_rows': (_i32, [_vp, _i64, _i64, _vp, _i64, _vp, _vp]),
    'ampconv_feat_zscore_stats': (_i32, [_vp, _i64, _i64, _vp, _vp, _vp]),
    'ampconv_feat_sample_present': (_i32, [_vp, _i64, _i32, _i32, ctypes.c_uint64, _vp, _vp, _vp]),
    'ampconv_feat_build': (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    'ampconv_feat_table_grad': (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    'ampconv_proj_supported': (_i32, [_i32, _i32, _i32]),
    'ampconv_proj_weight_image_bytes': (_sz, [_i32, _i32, _i32]),
    'ampconv_proj_weight_image': (_i32, [_vp, _i64, _i64, _i32, _i32, _vp, _i32, _vp]),
    'ampconv_proj_weight_images': (_i32, [_i32, _vp, _i32, _vp]),
    'ampconv_proj_rows': (_i32, [_vp, _i64, _i64, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _vp]),
    'ampconv_proj_wgrad_workspace_bytes': (_sz, [_i64, _i32, _i32, _i32]),
    'ampconv_proj_wgrad': (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _sz, _vp, _i64, _vp, _vp, _i32, _vp]),
    'ampconv_absmax': (_i32, [_vp, _i64, _i64, _i32, _i32, _vp, _i32, _vp]),
    'ampconv_planes_supported': (_i32, [_i32, _i32, _i32]),
    'ampconv_fwd_edge_planes': (_i32, [View, View, View, _vp, _vp, _i64, _i32, _i32, _i32, View, _vp, _i64, _vp, _vp, _vp]),
    'ampconv_bwd_edge_dst_planes': (_i32, [View, View, View, View, _vp, _vp, _i64, _i32, _i32, _i32, View, _vp, _i64, _vp,
                                           _vp, _vp, _vp, _vp, _vp]),
    'ampconv_bwd_edge_src_planes': (_i32, [View, View, View, View, _vp, _vp, _i64, _i32, _i32, _i32, View, View, _vp, _i64,
                                           _vp, _vp, _vp, _vp, _vp]),
    'ampconv_scaled_supported': (_i32, [_i32, _i32, _i32]),
    'ampconv_fwd_edge_scaled': (_i32, [View, View, View, _vp, _vp, _i64, _i32, _i32, _i32, View, _vp, _i64, _vp, _vp, _vp]),
    'ampconv_bwd_edge_dst_scaled': (_i32, [View, View, View, View, _vp, _vp, _i64, _i32, _i32, _i32, View, _vp, _i64, _vp,
                                           _vp, _vp, _vp, _vp, _vp]),
    'ampconv_bwd_edge_src_scaled': (_i32, [View, View, View, View, _vp, _vp, _vp, _i64, _i32, _i32, _i32, View, View, _vp,
                                           _i64, _vp, _vp, _vp, _vp, _vp]),
    'ampconv_proj_out_bound': (_i32, [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    'ampconv_proj_rows_planes': (_i32, [_vp, _i64, _i64, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp, _vp,
                                        _i32, _i32, _vp]),
    'ampconv_planes_to_f32': (_i32, [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _i64, _vp]),
    'ampconv_absmax_stats': (_i32, [_vp, _i64, _i64, _i32, _vp, _vp]),
    'ampconv_active_nodes_workspace_bytes': (_sz, [_i64]),
    'ampconv_active_nodes': (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    'ampconv_segment_mean': (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    'ampconv_mask_rows': (_i32, [_vp, _vp, _i64, _i64, _i32, _vp]),
    'ampconv_masked_colsum': (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _i32, _vp]),
}

_lib = None


class AmpconvError(RuntimeError):
    pass


def load():
    """Load libampconv.so once; raise (never fall back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AmpconvError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950). ampnet_amd has no non-HIP fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.ampconv_version()
    if got != EXPECTED_ABI:
        # argument lists differ between ABI versions (101 inserted the statistics pointers): a stale
        # or foreign build would reinterpret pointers as dtype / stream arguments
        raise AmpconvError(f'{LIB_PATH} has ABI version {got}, this binding needs {EXPECTED_ABI}: rebuild it '
                           '(python -c "import __graft_entry__ as g; g.build(force=True)")')
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ampconv_error_string(rc)
        raise AmpconvError(f'{what} failed: {msg.decode() if msg else rc} (code {rc})')
