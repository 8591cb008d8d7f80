// C-ABI entry points of the edge phase: argument checks + dispatch between the
// MFMA fast path (edge_mfma.hip) and the shape-generic kernels
// (edge_generic.hip).  AMPCONV_FORCE_GENERIC=1 in the environment pins the
// generic kernels (used by the tests to cross-check the two paths).
#include <cstdlib>
#include "common.h"

namespace {
bool force_generic() {
  const char *e = std::getenv("AMPCONV_FORCE_GENERIC");
  return e && e[0] == '1';
}
// short token sequences (L <= 4) have a family of their own (edge_small.hip); AMPCONV_SMALL=0 sends them to the tile
// kernels instead (the tests cross-check the two)
bool small_off() {
  const char *e = std::getenv("AMPCONV_SMALL");
  return e && e[0] == '0';
}
int check_common(int L, int D, int H, int dtype) {
  if (dtype != AMPCONV_F32 && dtype != AMPCONV_BF16) return AMPCONV_E_DTYPE;
  if (L <= 0 || D <= 0 || H <= 0 || D % H != 0) return AMPCONV_E_BADARG;
  return AMPCONV_OK;
}
// partial-tile view of the hub workspace: chunk c, token l, channel cc at P[(c*L + l)*D + cc]
ampconv_view_t partial_view(void *ws, int64_t tile, int64_t n_chunks, int L, int D, int H) {
  return ampconv_view_t{(float *)ws + tile * n_chunks * L * D, (int64_t)L * D, (int64_t)D,
                        (int64_t)(D / H)};
}
HubArgs hub_args(const void *plan, int mode) { return HubArgs{(const int32_t *)plan, mode}; }

}  // namespace

extern "C" int ampconv_version(void) { return AMPCONV_VERSION; }

extern "C" const char *ampconv_error_string(int code) {
  switch (code) {
    case AMPCONV_OK: return "ok";
    case AMPCONV_E_BADARG: return "ampconv: bad argument";
    case AMPCONV_E_DTYPE: return "ampconv: dtype not supported";
    case AMPCONV_E_WORKSPACE: return "ampconv: workspace too small";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "ampconv: unknown error";
  }
}

extern "C" int ampconv_fwd_edge(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                                const int32_t *rowptr, const int32_t *col, const int32_t *qidx,
                                int64_t n_rows, int L, int D, int H, ampconv_view_t O,
                                const void *hub_plan, int64_t hub_chunks, void *hub_ws, int dtype,
                                void *stream) {
  if (int rc = check_common(L, D, H, dtype)) return rc;
  if (n_rows < 0) return AMPCONV_E_BADARG;
  if (n_rows == 0) return AMPCONV_OK;
  if (!view_ok(Q) || !view_ok(K) || !view_ok(V) || !view_ok(O) || !rowptr) return AMPCONV_E_BADARG;
  const ampconv_view_t views[] = {Q, K, V, O};
  const bool bf = dtype == AMPCONV_BF16;
  if (bf && ampconv_bf16_supported(L, D, H, views, 4)) {
    hipStream_t st = (hipStream_t)stream;
    if (hub_plan && hub_chunks > 0 && hub_ws && !qidx) {
      if (int rc = ampconv_fwd_edge_bf16(Q, K, V, rowptr, col, nullptr, n_rows, L, D, H, O,
                                         hub_args(hub_plan, 1), st))
        return rc;
      ampconv_view_t P = partial_view(hub_ws, 0, hub_chunks, L, D, H);
      if (int rc = ampconv_fwd_edge_bf16(Q, K, V, rowptr, col, nullptr, hub_chunks, L, D, H, P,
                                         hub_args(hub_plan, 2), st))
        return rc;
      return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)P.ptr, O, rowptr, L, D, H, 1.f, 1, st);
    }
    return ampconv_fwd_edge_bf16(Q, K, V, rowptr, col, qidx, n_rows, L, D, H, O, HubArgs{nullptr, 0}, st);
  }
  // bf16 storage of the other shapes: the workgroup-per-unit kernels widen / round the rows themselves
  if (bf && !ampconv_block_supported(L, D, H, views, 4, true)) return AMPCONV_E_DTYPE;
  if (!bf && !force_generic() && !small_off() && ampconv_small_supported(L, D, H, views, 4)) {
    hipStream_t st = (hipStream_t)stream;
    if (hub_plan && hub_chunks > 0 && hub_ws && !qidx) {      // long segments: main + hub + combine
      if (int rc = ampconv_fwd_edge_small(Q, K, V, rowptr, col, nullptr, n_rows, L, D, H, O, hub_args(hub_plan, 1), st))
        return rc;
      ampconv_view_t P = partial_view(hub_ws, 0, hub_chunks, L, D, H);
      if (int rc = ampconv_fwd_edge_small(Q, K, V, rowptr, col, nullptr, hub_chunks, L, D, H, P, hub_args(hub_plan, 2), st))
        return rc;
      return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)P.ptr, O, rowptr, L, D, H, 1.f, 0, st);
    }
    return ampconv_fwd_edge_small(Q, K, V, rowptr, col, qidx, n_rows, L, D, H, O, HubArgs{nullptr, 0}, st);
  }
  if (!bf && !force_generic() && ampconv_mfma_supported(L, D, H) && ampconv_mfma_views_ok(views, 4)) {
    if (hub_plan && hub_chunks > 0 && hub_ws && !qidx) {      // long segments: main + hub + combine
      HubArgs hm = hub_args(hub_plan, 1), hh = hub_args(hub_plan, 2);
      if (int rc = ampconv_fwd_edge_mfma(Q, K, V, rowptr, col, nullptr, n_rows, L, D, H, O, hm,
                                         (hipStream_t)stream))
        return rc;
      ampconv_view_t P = partial_view(hub_ws, 0, hub_chunks, L, D, H);
      if (int rc = ampconv_fwd_edge_mfma(Q, K, V, rowptr, col, nullptr, hub_chunks, L, D, H, P, hh,
                                         (hipStream_t)stream))
        return rc;
      return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)P.ptr, O, rowptr, L, D, H, 1.f, 0,
                                 (hipStream_t)stream);
    }
    return ampconv_fwd_edge_mfma(Q, K, V, rowptr, col, qidx, n_rows, L, D, H, O, HubArgs{nullptr, 0},
                                 (hipStream_t)stream);
  }
  if ((bf || !force_generic()) && ampconv_block_supported(L, D, H, views, 4, bf)) {
    hipStream_t st = (hipStream_t)stream;
    if (hub_plan && hub_chunks > 0 && hub_ws && !qidx) {      // long segments: main + hub + combine
      if (int rc = ampconv_fwd_edge_block(Q, K, V, rowptr, col, nullptr, n_rows, L, D, H, O, hub_args(hub_plan, 1), bf, st))
        return rc;
      ampconv_view_t P = partial_view(hub_ws, 0, hub_chunks, L, D, H);
      if (int rc = ampconv_fwd_edge_block(Q, K, V, rowptr, col, nullptr, hub_chunks, L, D, H, P,
                                          hub_args(hub_plan, 2), bf, st))
        return rc;
      return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)P.ptr, O, rowptr, L, D, H, 1.f, bf ? 1 : 0, st);
    }
    return ampconv_fwd_edge_block(Q, K, V, rowptr, col, qidx, n_rows, L, D, H, O, HubArgs{nullptr, 0}, bf, st);
  }
  return ampconv_fwd_edge_generic(Q, K, V, rowptr, col, qidx, n_rows, L, D, H, O,
                                  (hipStream_t)stream);
}

extern "C" size_t ampconv_softmax_stats_bytes(int64_t E, int L, int D, int H, int dtype) {
  // bf16 storage keeps none: there the passes are HBM-bound and the extra 2 x 160 B per edge and
  // head cost the destination pass what they save the source pass (measured: +0.54 / -0.53 ms)
  if (E <= 0 || check_common(L, D, H, dtype) != AMPCONV_OK) return 0;
  if (dtype == AMPCONV_BF16)       // the bf16 MFMA kernels keep none; bf16 storage of the other shapes runs the
    return ampconv_bf16_supported(L, D, H, nullptr, 0) || !ampconv_block_supported(L, D, H, nullptr, 0, true)
               ? 0 : (size_t)E * H * ampconv_block_stats_floats(L) * sizeof(float);   // workgroup-per-unit kernels
  if (force_generic()) return 0;
  if (!small_off() && ampconv_small_supported(L, D, H, nullptr, 0)) return 0;      // edge_small.hip keeps none
  if (ampconv_mfma_supported(L, D, H))
    return (size_t)E * H * kStatsPerUnit * sizeof(float);
  if (ampconv_block_supported(L, D, H, nullptr, 0, false))     // shapes of the workgroup-per-unit kernels
    return (size_t)E * H * ampconv_block_stats_floats(L) * sizeof(float);
  return 0;
}

// ampconv_absmax (proj_gemm.hip)
extern "C" int ampconv_absmax(const void *X, int64_t ld, int64_t M, int K, int dtype, float *out, int reset, void *stream);

namespace {
// out_absmax of the backward passes for the kernel families that do not record it themselves: one pass over what was
// just written, which must then be a row-major matrix (token rows of D contiguous channels, nodes L rows apart)
int view_absmax(const ampconv_view_t &v, int64_t n, int L, int D, int H, float *out, void *stream) {
  if (v.head_stride != D / H || v.node_stride != (int64_t)L * v.row_stride) return AMPCONV_E_BADARG;
  return ampconv_absmax(v.ptr, v.row_stride, n * L, D, AMPCONV_F32, out, 0, stream);
}

int bwd_dst_dispatch(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dObar, const int32_t *rowptr,
                     const int32_t *col, int64_t n_rows, int L, int D, int H, ampconv_view_t dQ, const void *hub_plan,
                     int64_t hub_chunks, void *hub_ws, const int32_t *spos, float *stats, float *out_absmax,
                     bool *recorded, int dtype, void *stream) {
  if (int rc = check_common(L, D, H, dtype)) return rc;
  if (stats && (!spos || (uintptr_t)stats % 16 != 0)) return AMPCONV_E_BADARG;
  const StatsArgs sa{spos, stats, out_absmax};
  if (n_rows < 0) return AMPCONV_E_BADARG;
  if (n_rows == 0) return AMPCONV_OK;
  if (!view_ok(Q) || !view_ok(K) || !view_ok(V) || !view_ok(dObar) || !view_ok(dQ) || !rowptr)
    return AMPCONV_E_BADARG;
  const ampconv_view_t views[] = {Q, K, V, dObar, dQ};
  const bool bf = dtype == AMPCONV_BF16;
  if (bf && ampconv_bf16_supported(L, D, H, views, 5)) {
    if (stats) return AMPCONV_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (hub_plan && hub_chunks > 0 && hub_ws) {
      if (int rc = ampconv_bwd_edge_dst_bf16(Q, K, V, dObar, rowptr, col, n_rows, L, D, H, dQ,
                                             hub_args(hub_plan, 1), st))
        return rc;
      ampconv_view_t P = partial_view(hub_ws, 0, hub_chunks, L, D, H);
      if (int rc = ampconv_bwd_edge_dst_bf16(Q, K, V, dObar, rowptr, col, hub_chunks, L, D, H, P,
                                             hub_args(hub_plan, 2), st))
        return rc;
      return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)P.ptr, dQ, nullptr, L, D, H,
                                 1.f / sqrtf((float)(D / H)), 1, st);
    }
    return ampconv_bwd_edge_dst_bf16(Q, K, V, dObar, rowptr, col, n_rows, L, D, H, dQ, HubArgs{nullptr, 0}, st);
  }
  if (bf && !ampconv_block_supported(L, D, H, views, 5, true)) return AMPCONV_E_DTYPE;
  if (!bf && !stats && !force_generic() && !small_off() && ampconv_small_supported(L, D, H, views, 5)) {
    hipStream_t st = (hipStream_t)stream;
    if (hub_plan && hub_chunks > 0 && hub_ws) {
      if (int rc = ampconv_bwd_edge_dst_small(Q, K, V, dObar, rowptr, col, n_rows, L, D, H, dQ, hub_args(hub_plan, 1), st))
        return rc;
      ampconv_view_t P = partial_view(hub_ws, 0, hub_chunks, L, D, H);
      if (int rc = ampconv_bwd_edge_dst_small(Q, K, V, dObar, rowptr, col, hub_chunks, L, D, H, P, hub_args(hub_plan, 2), st))
        return rc;
      return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)P.ptr, dQ, nullptr, L, D, H,
                                 1.f / sqrtf((float)(D / H)), 0, st);
    }
    return ampconv_bwd_edge_dst_small(Q, K, V, dObar, rowptr, col, n_rows, L, D, H, dQ, HubArgs{nullptr, 0}, st);
  }
  if (!bf && !force_generic() && ampconv_mfma_supported(L, D, H) && ampconv_mfma_views_ok(views, 5)) {
    *recorded = true;          // these kernels (and the combine pass behind them) keep out_absmax themselves
    if (hub_plan && hub_chunks > 0 && hub_ws) {
      HubArgs hm = hub_args(hub_plan, 1), hh = hub_args(hub_plan, 2);
      if (int rc = ampconv_bwd_edge_dst_mfma(Q, K, V, dObar, rowptr, col, n_rows, L, D, H, dQ, hm, sa,
                                             (hipStream_t)stream))
        return rc;
      ampconv_view_t P = partial_view(hub_ws, 0, hub_chunks, L, D, H);
      if (int rc = ampconv_bwd_edge_dst_mfma(Q, K, V, dObar, rowptr, col, hub_chunks, L, D, H, P, hh,
                                             StatsArgs{sa.spos, sa.stats, nullptr},      // partial tiles: not recorded
                                             (hipStream_t)stream))
        return rc;
      return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)P.ptr, dQ, nullptr, L, D, H,
                                 1.f / sqrtf((float)(D / H)), 0, (hipStream_t)stream, out_absmax);
    }
    return ampconv_bwd_edge_dst_mfma(Q, K, V, dObar, rowptr, col, n_rows, L, D, H, dQ,
                                     HubArgs{nullptr, 0}, sa, (hipStream_t)stream);
  }
  // (a statistics buffer sized for the edge_mfma layout must not reach these kernels)
  if ((bf || !force_generic()) && ampconv_block_supported(L, D, H, views, 5, bf) &&
      !(stats && !bf && ampconv_mfma_supported(L, D, H))) {
    hipStream_t st = (hipStream_t)stream;
    if (hub_plan && hub_chunks > 0 && hub_ws) {
      if (int rc = ampconv_bwd_edge_dst_block(Q, K, V, dObar, rowptr, col, n_rows, L, D, H, dQ,
                                              hub_args(hub_plan, 1), sa, bf, st))
        return rc;
      ampconv_view_t P = partial_view(hub_ws, 0, hub_chunks, L, D, H);
      if (int rc = ampconv_bwd_edge_dst_block(Q, K, V, dObar, rowptr, col, hub_chunks, L, D, H, P,
                                              hub_args(hub_plan, 2), sa, bf, st))
        return rc;
      return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)P.ptr, dQ, nullptr, L, D, H,
                                 1.f / sqrtf((float)(D / H)), bf ? 1 : 0, st);
    }
    return ampconv_bwd_edge_dst_block(Q, K, V, dObar, rowptr, col, n_rows, L, D, H, dQ, HubArgs{nullptr, 0}, sa, bf, st);
  }
  if (bf) return AMPCONV_E_DTYPE;
  if (stats) return AMPCONV_E_BADARG;     // this shape's kernels keep no statistics (ampconv_softmax_stats_bytes = 0)
  return ampconv_bwd_edge_dst_generic(Q, K, V, dObar, rowptr, col, n_rows, L, D, H, dQ,
                                      (hipStream_t)stream);
}
}  // namespace

extern "C" int ampconv_bwd_edge_dst(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                                    ampconv_view_t dObar, const int32_t *rowptr,
                                    const int32_t *col, int64_t n_rows, int L, int D, int H,
                                    ampconv_view_t dQ, const void *hub_plan, int64_t hub_chunks,
                                    void *hub_ws, const int32_t *spos, float *stats, float *out_absmax,
                                    int dtype, void *stream) {
  if (out_absmax && dtype != AMPCONV_F32) return AMPCONV_E_DTYPE;       // operand maxima: fp32 storage (scaled projections)
  bool recorded = false;
  const int rc = bwd_dst_dispatch(Q, K, V, dObar, rowptr, col, n_rows, L, D, H, dQ, hub_plan, hub_chunks, hub_ws, spos,
                                  stats, out_absmax, &recorded, dtype, stream);
  if (rc != AMPCONV_OK || !out_absmax || recorded || n_rows == 0) return rc;
  return view_absmax(dQ, n_rows, L, D, H, out_absmax, stream);
}

namespace {
int bwd_src_dispatch(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dObar, const int32_t *cscptr,
                     const int32_t *crow, const float *cinv, int64_t n_src, int L, int D, int H, ampconv_view_t dK,
                     ampconv_view_t dV, const void *hub_plan, int64_t hub_chunks, void *hub_ws, const float *stats,
                     float *out_absmax, bool *recorded, int dtype, void *stream) {
  if (int rc = check_common(L, D, H, dtype)) return rc;
  if (stats && (uintptr_t)stats % 16 != 0) return AMPCONV_E_BADARG;
  const StatsArgs sa{nullptr, const_cast<float *>(stats), nullptr};
  if (n_src < 0) return AMPCONV_E_BADARG;
  if (n_src == 0) return AMPCONV_OK;
  if (!view_ok(Q) || !view_ok(K) || !view_ok(V) || !view_ok(dObar) || !view_ok(dK) ||
      !view_ok(dV) || !cscptr || !cinv)
    return AMPCONV_E_BADARG;
  const ampconv_view_t views[] = {Q, K, V, dObar, dK, dV};
  const bool bf = dtype == AMPCONV_BF16;
  if (bf && ampconv_bf16_supported(L, D, H, views, 6)) {
    if (stats) return AMPCONV_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (hub_plan && hub_chunks > 0 && hub_ws) {
      if (int rc = ampconv_bwd_edge_src_bf16(Q, K, V, dObar, cscptr, crow, cinv, n_src, L, D, H, dK, dV,
                                             hub_args(hub_plan, 1), st))
        return rc;
      ampconv_view_t PK = partial_view(hub_ws, 0, hub_chunks, L, D, H);
      ampconv_view_t PV = partial_view(hub_ws, 1, hub_chunks, L, D, H);
      if (int rc = ampconv_bwd_edge_src_bf16(Q, K, V, dObar, cscptr, crow, cinv, hub_chunks, L, D, H, PK,
                                             PV, hub_args(hub_plan, 2), st))
        return rc;
      if (int rc = ampconv_hub_combine(hub_plan, hub_chunks, (const float *)PK.ptr, dK, nullptr, L, D, H,
                                       1.f / sqrtf((float)(D / H)), 1, st))
        return rc;
      return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)PV.ptr, dV, nullptr, L, D, H, 1.f, 1, st);
    }
    return ampconv_bwd_edge_src_bf16(Q, K, V, dObar, cscptr, crow, cinv, n_src, L, D, H, dK, dV,
                                     HubArgs{nullptr, 0}, st);
  }
  if (bf && !(stats && ampconv_block_supported(L, D, H, views, 6, true))) return AMPCONV_E_DTYPE;
  if (!bf && !stats && !force_generic() && !small_off() && ampconv_small_supported(L, D, H, views, 6)) {
    hipStream_t st = (hipStream_t)stream;
    if (hub_plan && hub_chunks > 0 && hub_ws) {
      if (int rc = ampconv_bwd_edge_src_small(Q, K, V, dObar, cscptr, crow, cinv, n_src, L, D, H, dK, dV,
                                              hub_args(hub_plan, 1), st))
        return rc;
      ampconv_view_t PK = partial_view(hub_ws, 0, hub_chunks, L, D, H);
      ampconv_view_t PV = partial_view(hub_ws, 1, hub_chunks, L, D, H);
      if (int rc = ampconv_bwd_edge_src_small(Q, K, V, dObar, cscptr, crow, cinv, hub_chunks, L, D, H, PK, PV,
                                              hub_args(hub_plan, 2), st))
        return rc;
      if (int rc = ampconv_hub_combine(hub_plan, hub_chunks, (const float *)PK.ptr, dK, nullptr, L, D, H,
                                       0.6931471805599453f, 0, st))
        return rc;
      return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)PV.ptr, dV, nullptr, L, D, H, 1.f, 0, st);
    }
    return ampconv_bwd_edge_src_small(Q, K, V, dObar, cscptr, crow, cinv, n_src, L, D, H, dK, dV, HubArgs{nullptr, 0}, st);
  }
  if (!bf && !force_generic() && ampconv_mfma_supported(L, D, H) && ampconv_mfma_views_ok(views, 6)) {
    // (the source-pass kernels do not record out_absmax -- register budget, edge_mfma.hip: the pass below does)
    if (hub_plan && hub_chunks > 0 && hub_ws) {
      HubArgs hm = hub_args(hub_plan, 1), hh = hub_args(hub_plan, 2);
      if (int rc = ampconv_bwd_edge_src_mfma(Q, K, V, dObar, cscptr, crow, cinv, n_src, L, D, H, dK, dV,
                                             hm, sa, (hipStream_t)stream))
        return rc;
      ampconv_view_t PK = partial_view(hub_ws, 0, hub_chunks, L, D, H);
      ampconv_view_t PV = partial_view(hub_ws, 1, hub_chunks, L, D, H);
      if (int rc = ampconv_bwd_edge_src_mfma(Q, K, V, dObar, cscptr, crow, cinv, hub_chunks, L, D, H, PK,
                                             PV, hh, sa, (hipStream_t)stream))
        return rc;
      if (int rc = ampconv_hub_combine(hub_plan, hub_chunks, (const float *)PK.ptr, dK, nullptr, L, D, H,
                                       0.6931471805599453f, 0, (hipStream_t)stream))
        return rc;
      return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)PV.ptr, dV, nullptr, L, D, H, 1.f, 0,
                                 (hipStream_t)stream);
    }
    return ampconv_bwd_edge_src_mfma(Q, K, V, dObar, cscptr, crow, cinv, n_src, L, D, H, dK, dV,
                                     HubArgs{nullptr, 0}, sa, (hipStream_t)stream);
  }
  // the workgroup-per-unit source pass exists only with the statistics; without them: generic kernels
  if (stats && (bf || !force_generic()) && ampconv_block_supported(L, D, H, views, 6, bf) &&
      (bf || !ampconv_mfma_supported(L, D, H))) {
    hipStream_t st = (hipStream_t)stream;
    const int obf = bf ? 1 : 0;
    if (hub_plan && hub_chunks > 0 && hub_ws) {
      if (int rc = ampconv_bwd_edge_src_block(Q, K, V, dObar, cscptr, crow, cinv, n_src, L, D, H, dK, dV,
                                              hub_args(hub_plan, 1), stats, bf, st))
        return rc;
      ampconv_view_t PK = partial_view(hub_ws, 0, hub_chunks, L, D, H);
      ampconv_view_t PV = partial_view(hub_ws, 1, hub_chunks, L, D, H);
      if (int rc = ampconv_bwd_edge_src_block(Q, K, V, dObar, cscptr, crow, cinv, hub_chunks, L, D, H, PK, PV,
                                              hub_args(hub_plan, 2), stats, bf, st))
        return rc;
      if (int rc = ampconv_hub_combine(hub_plan, hub_chunks, (const float *)PK.ptr, dK, nullptr, L, D, H,
                                       0.6931471805599453f, obf, st))
        return rc;
      return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)PV.ptr, dV, nullptr, L, D, H, 1.f, obf, st);
    }
    return ampconv_bwd_edge_src_block(Q, K, V, dObar, cscptr, crow, cinv, n_src, L, D, H, dK, dV,
                                      HubArgs{nullptr, 0}, stats, bf, st);
  }
  if (bf) return AMPCONV_E_DTYPE;
  if (stats) return AMPCONV_E_BADARG;
  return ampconv_bwd_edge_src_generic(Q, K, V, dObar, cscptr, crow, cinv, n_src, L, D, H, dK,
                                      dV, (hipStream_t)stream);
}
}  // namespace

extern "C" int ampconv_bwd_edge_src(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                                    ampconv_view_t dObar, const int32_t *cscptr,
                                    const int32_t *crow, const float *cinv, int64_t n_src,
                                    int L, int D, int H, ampconv_view_t dK, ampconv_view_t dV,
                                    const void *hub_plan, int64_t hub_chunks, void *hub_ws,
                                    const float *stats, float *out_absmax, int dtype, void *stream) {
  if (out_absmax && dtype != AMPCONV_F32) return AMPCONV_E_DTYPE;
  bool recorded = false;
  const int rc = bwd_src_dispatch(Q, K, V, dObar, cscptr, crow, cinv, n_src, L, D, H, dK, dV, hub_plan, hub_chunks, hub_ws,
                                  stats, out_absmax, &recorded, dtype, stream);
  if (rc != AMPCONV_OK || !out_absmax || recorded || n_src == 0) return rc;
  if (int rc2 = view_absmax(dK, n_src, L, D, H, out_absmax, stream)) return rc2;
  return view_absmax(dV, n_src, L, D, H, out_absmax, stream);
}
