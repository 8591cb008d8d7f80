// Edge-phase kernels for SHORT token sequences: L <= 4 tokens per node (the L = 1 and L = 4 sweeps of
// BASELINE config 3; the reference's own toy harness runs L = 2).  The MFMA families pad every node to 16- or 20-token
// tiles: at L = 1 they spend a 20 x 20 score tile on one score (measured, round 4: 4.4 / 7.0 / 6.7 ms for 1 M edges at
// L = 1, D = 128 -- the time of L = 20 -- against 0.7 ms of HBM time for all three passes).  Here nothing is padded:
//
//   one wavefront owns one ROW (all heads); lane l holds VPL consecutive channels of every token row (VPL = 1, 2 or 4 so
//   that D / VPL <= 64), a head is LPH = dh / VPL consecutive lanes (a power of two), a streamed tile is L x D floats =
//   L vector loads per lane of whole contiguous rows, the L x L scores of a head are per-lane partial dot products
//   all-reduced over the head's lanes (DPP butterflies inside a 16-lane row, a bpermute / permlane step beyond), softmax
//   and the weighted sums are in-lane VALU.  No LDS, no MFMA, no atomics; sums in CSR order (bitwise reproducible); long
//   segments through the same plan / partial-tile / combine passes as the other families (hub.hip).
//
// Reference arithmetic replaced: the same lines as edge_mfma.hip (torch functional.py:6578-6594, amp_conv.py:11);
// backward per SURVEY.md A.2.  The source pass re-derives the softmax itself (an L x L tile per head costs a few VALU
// instructions): no statistics are handed over for these shapes.
#include "mfma_tile.h"

namespace {

constexpr float kLog2eS = 1.4426950408889634f;
constexpr int kWavesPerBlockS = 4;

struct SArgs {
  ampconv_view_t Q, K, V, dO, O, dK, dV;     // O = forward output / dQ
  const int32_t *ptr, *idx, *qidx;
  const float *cinv;
  HubArgs hub;
  int64_t n_units;
  int D, dh;
  float qscale, oscale;
};

// all-reduce (sum / max) over the LPH consecutive lanes of a head
template <int LPH>
__device__ __forceinline__ float head_sum(float x) {
  if constexpr (LPH >= 2) x += dpp_mov<0xB1>(x);     // quad_perm [1,0,3,2]
  if constexpr (LPH >= 4) x += dpp_mov<0x4E>(x);     // quad_perm [2,3,0,1]
  if constexpr (LPH >= 8) x += dpp_mov<0x141>(x);    // row_half_mirror
  if constexpr (LPH >= 16) x += dpp_mov<0x140>(x);   // row_mirror
  if constexpr (LPH >= 32) x += __shfl_xor(x, 16, 64);
  if constexpr (LPH >= 64) x += __shfl_xor(x, 32, 64);
  return x;
}

template <int VPL>
struct Vec {
  float v[VPL];
};
template <int VPL>
__device__ __forceinline__ Vec<VPL> load_vec(const float *p) {
  Vec<VPL> r;
  if constexpr (VPL == 4) {
    const float4 x = *reinterpret_cast<const float4 *>(p);
    r.v[0] = x.x; r.v[1] = x.y; r.v[2] = x.z; r.v[3] = x.w;
  } else if constexpr (VPL == 2) {
    const float2 x = *reinterpret_cast<const float2 *>(p);
    r.v[0] = x.x; r.v[1] = x.y;
  } else {
    r.v[0] = *p;
  }
  return r;
}
template <int VPL>
__device__ __forceinline__ void store_vec(float *p, const Vec<VPL> &r) {
  if constexpr (VPL == 4) *reinterpret_cast<float4 *>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  else if constexpr (VPL == 2) *reinterpret_cast<float2 *>(p) = make_float2(r.v[0], r.v[1]);
  else *p = r.v[0];
}

// the L token rows of node n as this lane sees them (its VPL channels of every row); `on` = lane holds channels at all
template <int L, int VPL>
struct Tile {
  Vec<VPL> r[L];
};
template <int L, int VPL>
__device__ __forceinline__ void tile_load_s(Tile<L, VPL> &t, const ampconv_view_t &v, int64_t n, int64_t loff, bool on) {
  const float *base = reinterpret_cast<const float *>(v.ptr) + n * v.node_stride + loff;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    if (on) t.r[l] = load_vec<VPL>(base + (int64_t)l * v.row_stride);
    else
#pragma unroll
      for (int k = 0; k < VPL; ++k) t.r[l].v[k] = 0.f;
  }
}
template <int L, int VPL>
__device__ __forceinline__ void tile_store_s(const ampconv_view_t &v, int64_t n, int64_t loff, bool on, const Tile<L, VPL> &t,
                                             float scale) {
  if (!on) return;
  float *base = reinterpret_cast<float *>(v.ptr) + n * v.node_stride + loff;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Vec<VPL> o;
#pragma unroll
    for (int k = 0; k < VPL; ++k) o.v[k] = t.r[l].v[k] * scale;
    store_vec<VPL>(base + (int64_t)l * v.row_stride, o);
  }
}

// head-wise products of two tiles: out[i][j] = sum over the head's channels of a[i][c] b[j][c]
template <int L, int VPL, int LPH>
__device__ __forceinline__ void head_dots(float (&out)[L][L], const Tile<L, VPL> &a, const Tile<L, VPL> &b) {
#pragma unroll
  for (int i = 0; i < L; ++i)
#pragma unroll
    for (int j = 0; j < L; ++j) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < VPL; ++k) s = fmaf(a.r[i].v[k], b.r[j].v[k], s);
      out[i][j] = head_sum<LPH>(s);
    }
}
// row softmax of S (already in log2 units) in place; every lane of the head holds the whole L x L tile
template <int L>
__device__ __forceinline__ void softmax_rows(float (&S)[L][L]) {
#pragma unroll
  for (int i = 0; i < L; ++i) {
    float m = S[i][0];
#pragma unroll
    for (int j = 1; j < L; ++j) m = fmaxf(m, S[i][j]);
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < L; ++j) {
      S[i][j] = fast_exp2(S[i][j] - m);
      l += S[i][j];
    }
    const float inv = fast_rcp(l);
#pragma unroll
    for (int j = 0; j < L; ++j) S[i][j] *= inv;
  }
}

// lane geometry: channel offset of the lane inside a token row, through the view's head stride
struct LaneMap {
  int64_t qo, ko, vo, go, oo, dko, dvo;
  bool on;
};
template <int VPL>
__device__ __forceinline__ LaneMap lane_map(const SArgs &a, int lane) {
  LaneMap m;
  const int c0 = VPL * lane;
  m.on = c0 < a.D;
  const int h = m.on ? c0 / a.dh : 0, c = m.on ? c0 - h * a.dh : 0;
  auto off = [&](const ampconv_view_t &v) { return (int64_t)h * v.head_stride + c; };
  m.qo = off(a.Q); m.ko = off(a.K); m.vo = off(a.V); m.go = off(a.dO); m.oo = off(a.O); m.dko = off(a.dK); m.dvo = off(a.dV);
  return m;
}

// ---------------------------------------------------------------- forward
template <int L, int VPL, int LPH>
__global__ __launch_bounds__(64 * kWavesPerBlockS) void fwd_small(SArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlockS + wave;
  if (unit >= a.n_units) return;
  int64_t r, onode;
  int h1, beg, end, deg;
  if (!map_unit(a.hub, a.ptr, unit, a.n_units, 1, r, onode, h1, beg, end, deg)) return;
  const LaneMap lm = lane_map<VPL>(a, lane);
  const int64_t d = a.qidx ? a.qidx[r] : r;

  Tile<L, VPL> q, acc;
  tile_load_s<L, VPL>(q, a.Q, d, lm.qo, lm.on);
#pragma unroll
  for (int i = 0; i < L; ++i)
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      q.r[i].v[k] *= a.qscale;
      acc.r[i].v[k] = 0.f;
    }
  Tile<L, VPL> kn, vn;
  IdxWindow win;
  auto fetch = [&](int p) {
    const int64_t s = idxwin_get<false>(win, a.idx, nullptr, p, end, lane, nullptr);
    tile_load_s<L, VPL>(kn, a.K, s, lm.ko, lm.on);
    tile_load_s<L, VPL>(vn, a.V, s, lm.vo, lm.on);
  };
  if (beg < end) {
    idxwin_load<false>(win, a.idx, nullptr, beg, end, lane);
    fetch(beg);
  }
  for (int p = beg; p < end; ++p) {
    const Tile<L, VPL> k = kn, v = vn;
    if (p + 1 < end) fetch(p + 1);
    float S[L][L];
    head_dots<L, VPL, LPH>(S, q, k);
    softmax_rows<L>(S);
#pragma unroll
    for (int i = 0; i < L; ++i)
#pragma unroll
      for (int j = 0; j < L; ++j)
#pragma unroll
        for (int kk = 0; kk < VPL; ++kk) acc.r[i].v[kk] = fmaf(S[i][j], v.r[j].v[kk], acc.r[i].v[kk]);
  }
  // hub pass: unnormalised partial tile, the combine pass applies 1/deg
  tile_store_s<L, VPL>(a.O, onode, lm.oo, lm.on, acc, a.hub.mode == 2 ? 1.f : (deg > 0 ? 1.f / (float)deg : 0.f));
}

// ---------------------------------------------------------------- backward, destination pass: dQ
template <int L, int VPL, int LPH>
__global__ __launch_bounds__(64 * kWavesPerBlockS) void bwd_dst_small(SArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlockS + wave;
  if (unit >= a.n_units) return;
  int64_t r, onode;
  int h1, beg, end, deg;
  if (!map_unit(a.hub, a.ptr, unit, a.n_units, 1, r, onode, h1, beg, end, deg)) return;
  const LaneMap lm = lane_map<VPL>(a, lane);
  const float inv = deg > 0 ? 1.f / (float)deg : 0.f;       // dO is the gradient of the MEAN

  Tile<L, VPL> q, g, acc;
  tile_load_s<L, VPL>(q, a.Q, r, lm.qo, lm.on);
  tile_load_s<L, VPL>(g, a.dO, r, lm.go, lm.on);
#pragma unroll
  for (int i = 0; i < L; ++i)
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      q.r[i].v[k] *= a.qscale;
      g.r[i].v[k] *= inv;
      acc.r[i].v[k] = 0.f;
    }
  Tile<L, VPL> kn, vn;
  IdxWindow win;
  auto fetch = [&](int p) {
    const int64_t s = idxwin_get<false>(win, a.idx, nullptr, p, end, lane, nullptr);
    tile_load_s<L, VPL>(kn, a.K, s, lm.ko, lm.on);
    tile_load_s<L, VPL>(vn, a.V, s, lm.vo, lm.on);
  };
  if (beg < end) {
    idxwin_load<false>(win, a.idx, nullptr, beg, end, lane);
    fetch(beg);
  }
  for (int p = beg; p < end; ++p) {
    const Tile<L, VPL> k = kn, v = vn;
    if (p + 1 < end) fetch(p + 1);
    float S[L][L], dP[L][L];
    head_dots<L, VPL, LPH>(S, q, k);
    head_dots<L, VPL, LPH>(dP, g, v);
    softmax_rows<L>(S);
#pragma unroll
    for (int i = 0; i < L; ++i) {
      float delta = 0.f;
#pragma unroll
      for (int j = 0; j < L; ++j) delta = fmaf(S[i][j], dP[i][j], delta);
#pragma unroll
      for (int j = 0; j < L; ++j) {
        const float dS = S[i][j] * (dP[i][j] - delta);
#pragma unroll
        for (int kk = 0; kk < VPL; ++kk) acc.r[i].v[kk] = fmaf(dS, k.r[j].v[kk], acc.r[i].v[kk]);
      }
    }
  }
  tile_store_s<L, VPL>(a.O, onode, lm.oo, lm.on, acc, a.hub.mode == 2 ? 1.f : a.oscale);
}

// ---------------------------------------------------------------- backward, source pass: dK, dV
template <int L, int VPL, int LPH>
__global__ __launch_bounds__(64 * kWavesPerBlockS) void bwd_src_small(SArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlockS + wave;
  if (unit >= a.n_units) return;
  int64_t s, onode;
  int h1, beg, end, deg;
  if (!map_unit(a.hub, a.ptr, unit, a.n_units, 1, s, onode, h1, beg, end, deg)) return;
  const LaneMap lm = lane_map<VPL>(a, lane);

  Tile<L, VPL> k, v, dk, dv;
  tile_load_s<L, VPL>(k, a.K, s, lm.ko, lm.on);
  tile_load_s<L, VPL>(v, a.V, s, lm.vo, lm.on);
#pragma unroll
  for (int i = 0; i < L; ++i)
#pragma unroll
    for (int kk = 0; kk < VPL; ++kk) dk.r[i].v[kk] = dv.r[i].v[kk] = 0.f;
  Tile<L, VPL> qn, gn;
  float inv_next = 0.f;
  IdxWindow win;
  auto fetch = [&](int p) {
    const int64_t d = idxwin_get<true>(win, a.idx, a.cinv, p, end, lane, &inv_next);
    tile_load_s<L, VPL>(qn, a.Q, d, lm.qo, lm.on);
    tile_load_s<L, VPL>(gn, a.dO, d, lm.go, lm.on);
  };
  if (beg < end) {
    idxwin_load<true>(win, a.idx, a.cinv, beg, end, lane);
    fetch(beg);
  }
  for (int p = beg; p < end; ++p) {
    Tile<L, VPL> q = qn, g = gn;
    const float inv = inv_next;
    if (p + 1 < end) fetch(p + 1);
#pragma unroll
    for (int i = 0; i < L; ++i)
#pragma unroll
      for (int kk = 0; kk < VPL; ++kk) {
        q.r[i].v[kk] *= a.qscale;
        g.r[i].v[kk] *= inv;
      }
    float S[L][L], dP[L][L];          // [destination token i][source token j]
    head_dots<L, VPL, LPH>(S, q, k);
    head_dots<L, VPL, LPH>(dP, g, v);
    softmax_rows<L>(S);
#pragma unroll
    for (int i = 0; i < L; ++i) {
      float delta = 0.f;
#pragma unroll
      for (int j = 0; j < L; ++j) delta = fmaf(S[i][j], dP[i][j], delta);
#pragma unroll
      for (int j = 0; j < L; ++j) {
        const float dS = S[i][j] * (dP[i][j] - delta);
#pragma unroll
        for (int kk = 0; kk < VPL; ++kk) {
          dv.r[j].v[kk] = fmaf(S[i][j], g.r[i].v[kk], dv.r[j].v[kk]);
          dk.r[j].v[kk] = fmaf(dS, q.r[i].v[kk], dk.r[j].v[kk]);
        }
      }
    }
  }
  // q carried log2e / sqrt(dh): dK = ln2 * sum dS^T q  (the factor the other families' source pass applies)
  tile_store_s<L, VPL>(a.dK, onode, lm.dko, lm.on, dk, a.hub.mode == 2 ? 1.f : a.oscale);
  tile_store_s<L, VPL>(a.dV, onode, lm.dvo, lm.on, dv, 1.f);
}

// ---- dispatch over (L, VPL, LPH)
struct SmallShape {
  int vpl, lph;
};
inline bool small_shape(int L, int D, int H, SmallShape &sh) {
  if (L < 1 || L > 4 || D <= 0 || H <= 0 || D % H) return false;
  const int dh = D / H;
  for (int vpl = 1; vpl <= 4; vpl *= 2) {
    if (D % vpl || dh % vpl || D / vpl > 64) continue;
    const int lph = dh / vpl;
    if (lph != 4 && lph != 8 && lph != 16 && lph != 32) continue;     // a head = a power-of-two group of 4 .. 32 lanes
    sh.vpl = vpl;
    sh.lph = lph;
    return true;
  }
  return false;
}

typedef void (*SmallKernel)(SArgs);
template <template <int, int, int> class F, int L, int VPL>
SmallKernel by_lph(int lph) {
  switch (lph) {
    case 4: return F<L, VPL, 4>::get();
    case 8: return F<L, VPL, 8>::get();
    case 16: return F<L, VPL, 16>::get();
    default: return F<L, VPL, 32>::get();
  }
}
template <template <int, int, int> class F, int L>
SmallKernel by_vpl(int vpl, int lph) {
  return vpl == 4 ? by_lph<F, L, 4>(lph) : vpl == 2 ? by_lph<F, L, 2>(lph) : by_lph<F, L, 1>(lph);
}
template <template <int, int, int> class F>
SmallKernel pick(int L, int vpl, int lph) {
  switch (L) {
    case 1: return by_vpl<F, 1>(vpl, lph);
    case 2: return by_vpl<F, 2>(vpl, lph);
    case 3: return by_vpl<F, 3>(vpl, lph);
    default: return by_vpl<F, 4>(vpl, lph);
  }
}
template <int L, int VPL, int LPH> struct FwdS { static SmallKernel get() { return fwd_small<L, VPL, LPH>; } };
template <int L, int VPL, int LPH> struct DstS { static SmallKernel get() { return bwd_dst_small<L, VPL, LPH>; } };
template <int L, int VPL, int LPH> struct SrcS { static SmallKernel get() { return bwd_src_small<L, VPL, LPH>; } };

int launch_small(SArgs &a, SmallKernel k, int D, int H, hipStream_t stream) {
  a.D = D;
  a.dh = D / H;
  a.qscale = kLog2eS / sqrtf((float)a.dh);
  const int64_t blocks = (a.n_units + kWavesPerBlockS - 1) / kWavesPerBlockS;
  if (blocks > INT32_MAX) return AMPCONV_E_BADARG;
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * kWavesPerBlockS), 0, stream, a);
  return ampconv_launch_status();
}

}  // namespace

bool ampconv_small_supported(int L, int D, int H, const ampconv_view_t *views, int n) {
  SmallShape sh;
  if (!small_shape(L, D, H, sh)) return false;
  for (int i = 0; i < n; ++i) {
    const ampconv_view_t &v = views[i];
    if ((uintptr_t)v.ptr % (4 * sh.vpl) || v.node_stride % sh.vpl || v.row_stride % sh.vpl || v.head_stride % sh.vpl)
      return false;
  }
  return true;
}

int ampconv_fwd_edge_small(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, const int32_t *rowptr,
                           const int32_t *col, const int32_t *qidx, int64_t n_rows, int L, int D, int H,
                           ampconv_view_t O, HubArgs hub, hipStream_t stream) {
  SmallShape sh;
  if (!small_shape(L, D, H, sh)) return AMPCONV_E_BADARG;
  SArgs a{};
  a.Q = Q; a.K = K; a.V = V; a.O = O; a.dO = Q; a.dK = Q; a.dV = Q;
  a.ptr = rowptr; a.idx = col; a.qidx = qidx; a.hub = hub;
  a.n_units = n_rows;
  return launch_small(a, pick<FwdS>(L, sh.vpl, sh.lph), D, H, stream);
}

int ampconv_bwd_edge_dst_small(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dO,
                               const int32_t *rowptr, const int32_t *col, int64_t n_rows, int L, int D, int H,
                               ampconv_view_t dQ, HubArgs hub, hipStream_t stream) {
  SmallShape sh;
  if (!small_shape(L, D, H, sh)) return AMPCONV_E_BADARG;
  SArgs a{};
  a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.O = dQ; a.dK = Q; a.dV = Q;
  a.ptr = rowptr; a.idx = col; a.hub = hub;
  a.n_units = n_rows;
  a.oscale = 1.f / sqrtf((float)(D / H));
  return launch_small(a, pick<DstS>(L, sh.vpl, sh.lph), D, H, stream);
}

int ampconv_bwd_edge_src_small(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dO,
                               const int32_t *cscptr, const int32_t *crow, const float *cinv, int64_t n_src,
                               int L, int D, int H, ampconv_view_t dK, ampconv_view_t dV, HubArgs hub,
                               hipStream_t stream) {
  SmallShape sh;
  if (!small_shape(L, D, H, sh)) return AMPCONV_E_BADARG;
  SArgs a{};
  a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.O = Q; a.dK = dK; a.dV = dV;
  a.ptr = cscptr; a.idx = crow; a.cinv = cinv; a.hub = hub;
  a.n_units = n_src;
  a.oscale = 0.6931471805599453f;       // dK = ln2 * sum dS^T (Q * log2e / sqrt(dh))
  return launch_small(a, pick<SrcS>(L, sh.vpl, sh.lph), D, H, stream);
}
