// fp32-grade edge kernels off the FP32 pipe (gfx950): Q/K/V and dObar arrive as TWO fp16 PLANES of the
// power-of-two-scaled fp32 value (written that way by the projection that produces them: proj_gemm.hip, PLANES
// epilogue), every product is the fp32 sum of three v_mfma_f32_16x16x32_f16 partial products
//      a b ~ a_lo b_hi + a_hi b_lo + a_hi b_hi        (dropped: a_lo b_lo <= 2^-22 |a b|)
// softmax, delta and all sums stay fp32, outputs (Obar, dQ, dK, dV) are plain fp32.  Same bytes per element as the
// fp32 path, 5 % of its matrix-pipe cycles per product (3 x 16 instead of 8 x 32 x 4 tiles), and the 16-bit matrix
// pipe runs beside the VALU instead of on it.
//
// Reference arithmetic replaced: torch functional.py:6578 (scale), :6589 (QK^T), :6590 (softmax), :6594 (PV) per edge,
// the mean aggregation of amp_conv.py:11, and their autograd backward.
//
// PLANE FORMAT ("f16x2", include/ampconv.h): the 128-byte slot of the 32 fp32 channels of one (token row, head) holds
//   bytes  0.. 63: hi[c] = fp16(x[c] * 2^e),               c = 0..31
//   bytes 64..127: lo[c] = fp16(x[c] * 2^e - hi[c])
// with ONE exponent e per tensor, e = 14 - floor(log2 bound) for a device-side bound of max |x| (plane_scale: the scaled
// maximum lies below 2^15).  Views keep the strides of the fp32 tensor the planes replace (in 4-byte elements).
//
// Structure = edge_mfma_bf16.hip with two plane images per tile: one wavefront owns one (row, head) unit, streams a
// pair of 20 x 128-byte tiles per edge (five full wave-wide 16-B/lane loads) through registers into private LDS
// images (per plane the swizzled 64-byte-row image of the bf16 kernels: conflict-free ds_write_b128, ds_read_b128 and
// transposed reads), channel-product fragments are one ds_read_b128 per plane, token-product fragments two
// ds_read_b64_tr_b16 per plane, softmax results are split into two fp16 planes in registers and are the B operand of
// the next product as they stand.  dObar arrives DIVIDED by the in-degree of its node (the producing projection's
// epilogue does it), so neither backward pass carries a per-edge weight.  Softmax statistics (optional, as in the fp32
// kernels: 20 log2-sum-exp + 20 delta floats per edge and head, filed at the edge's CSC position by the destination pass):
// with them the source pass's softmax is element-wise -- its three 16-lane reductions per row are a third of its
// vector instructions, and these passes are bound by vector issue and by the clock the chip holds under them.
#include "mfma_tile.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr float kLog2e = 1.4426950408889634f;
constexpr int kWavesPerBlock = 4;
constexpr int DH = 32;
// HALF (dh = 16, BASELINE config 3's head width): the head's slot is 64 bytes (16 hi, 16 lo); its 16 channels occupy the
// lower half of every 32-channel plane row, the upper half is never written (the images are zeroed once per unit: the
// 32-deep channel contraction is zero-padded) and only channel tile mc = 0 exists.  Half the bytes per (edge, head) at the
// same vector work: these shapes are bound by instruction issue, not by HBM.
constexpr int kRowBytes = DH * 2;                      // one plane of a token row
constexpr int kPlaneBytes = kLmax * kRowBytes;         // 1280: one plane image
// the lo image sits an odd multiple of 64 bytes behind the hi image: the 8 lanes that file one 128-byte row (4 hi
// chunks, 4 lo chunks) then cover all 32 store banks once
constexpr int kLoOff = kPlaneBytes + 64;               // 1344
constexpr int kTileBytes = kLoOff + kPlaneBytes;       // 2624 (a multiple of 16)
constexpr float kPScale = 16384.f;                     // softmax weights (<= 1) are split as P * 2^14
constexpr float kPUnscale = 1.f / 16384.f;
// dS = P (dP - delta), in the units of dP' = dO' V'^T, is split with ONE power-of-two scale per unit from a bound of
// what the unit can see: |dP'_ij| <= |dO'_i| |V'_j| (Cauchy-Schwarz over the 32 channels), |dP - delta| <= 2 max |dP|,
// P <= 1.  The unit's OWN side enters with the largest token-row norm it really has (computed once per unit from its
// hi plane), the streamed side with sqrt(32) x the recorded maximum of its tensor.  (The a-priori worst case --
// 32 channels at 2^15 each -- is 2^17 above typical data: scaled by it, dS sits at 2^-9 and its low plane is a
// subnormal with four bits left.  Measured: dx error 4e-6 of the maximum instead of 8e-7.)
constexpr float kSqrtDh = 5.656854249492381f;
// mask value of the softmaxes: the scores here are in the units of Q' K'^T and meet their scale only inside the
// exponential, so a large finite mask (kNegBig) could be scaled back into range; -inf stays -inf (every column / row
// has at least one finite score: token 0)
constexpr float kMasked = -__builtin_inff();

#define MFMA_F16(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), (c), 0, 0, 0)

// the three partial products of one fp32-grade product, smallest first
__device__ __forceinline__ f32x4 mfma3(const i32x4 &ah, const i32x4 &al, const i32x4 &bh, const i32x4 &bl, f32x4 c) {
  c = MFMA_F16(al, bh, c);
  c = MFMA_F16(ah, bl, c);
  return MFMA_F16(ah, bh, c);
}

// scale of a tensor whose magnitudes are bounded by `bound` (the function of proj_gemm.hip: both sides of the hand-off
// must derive the same power of two): 2^(14 - floor(log2 bound)), exponent field clamped to [15, 254]
__device__ __forceinline__ float plane_scale(float bound) {
  int e = (int)((__builtin_bit_cast(unsigned, bound) >> 23) & 0xFFu);
  e = e < 15 ? 15 : (e > 254 ? 254 : e);
  return __builtin_bit_cast(float, (unsigned)(268 - e) << 23);
}

__device__ __forceinline__ int cvt_pk_f16(float a, float b) {      // v_cvt_pk_f16_f32 (RNE)
  f32x2 v = {a, b};
  return __builtin_bit_cast(int, __builtin_convertvector(v, f16x2));
}
// (x0, x1) -> packed fp16 pairs of the two planes; |x| < 2^16
__device__ __forceinline__ void split2(float x0, float x1, int &h, int &l) {
  h = cvt_pk_f16(x0, x1);
  const f16x2 hv = __builtin_bit_cast(f16x2, h);
  l = cvt_pk_f16(x0 - (float)hv[0], x1 - (float)hv[1]);
}

// sum of squares of the 8 fp16 values of a fragment register quadruple (v_dot2_f32_f16)
__device__ __forceinline__ float frag_sumsq(const i32x4 &f) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f16x2 v = __builtin_bit_cast(f16x2, f[k]);
    s = __builtin_amdgcn_fdot2(v, v, s, false);
  }
  return s;
}
// largest token-row norm^2 of a tile given as its two channel-product fragments (lane (n, kg) = 8 channels of the token
// of column n): wave-uniform
__device__ __forceinline__ float tile_max_norm2(const i32x4 &f0, const i32x4 &f1) {
  const float t = fmaxf(groups_sum(frag_sumsq(f0)), groups_sum(frag_sumsq(f1)));
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, row16_max(t))));
}
// scale of dS for a unit whose own side has largest row norm^2 `own2` and whose streamed side is bounded by `other`
// per element (both in plane units): dS * scale < 2^15
__device__ __forceinline__ float ds_scale(float own2, float other) {
  return plane_scale(2.f * __builtin_sqrtf(own2) * (kSqrtDh * other));
}

// byte offset of 16-byte chunk `ch` (0..3) of token row `j` in a plane image (the swizzle of edge_mfma_bf16.hip)
__device__ __forceinline__ int plane_off(int j, int ch) {
  const int f = (4 - (j >> 2)) & 3;
  return j * kRowBytes + ((ch ^ f) << 4);
}

// ---- streamed tile pair: lane (r = lane >> 3, q = lane & 7) owns 16-byte chunk q (0..3: hi plane, 4..7: lo plane) of
// pair-rows r + 8 i, i < 5; pair-rows 0..19 = tile A, 20..39 = tile B.  Strides in bytes.
struct PairRegsP {
  i32x4 v[5];
};

template <bool FULL, bool HALF>
__device__ __forceinline__ void pair_load_p(PairRegsP &t, const char *baseA, unsigned strideA, const char *baseB,
                                            unsigned strideB, int L, int lane) {
  // dh = 32: 8 lanes per 128-byte slot row, 5 loads; HALF: 4 lanes per 64-byte slot row (chunks 0, 1 hi; 2, 3 lo), 3 loads
  constexpr int LPR = HALF ? 4 : 8, RPL = 64 / LPR, NLD = HALF ? 3 : 5;
  const int r = lane / LPR, q = lane % LPR;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int R = r + RPL * i;
    const bool isB = R >= kLmax;
    const int j = isB ? R - kLmax : R;
    // per-lane part as a 32-bit byte offset on top of a tile base that is uniform per edge (base stays in SGPRs)
    const unsigned off = (unsigned)j * (isB ? strideB : strideA) + 16u * (unsigned)q;
    const char *p = (isB ? baseB : baseA) + off;
    if (R < 2 * kLmax && (FULL || j < L)) t.v[i] = *reinterpret_cast<const i32x4 *>(p);
  }
}

template <bool FULL, bool HALF>
__device__ __forceinline__ void pair_to_lds_p(char *tileA, const PairRegsP &t, int L, int lane) {
  constexpr int LPR = HALF ? 4 : 8, RPL = 64 / LPR, NLD = HALF ? 3 : 5, CPP = LPR / 2;      // chunks per plane
  const int r = lane / LPR, q = lane % LPR;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int R = r + RPL * i;
    const bool isB = R >= kLmax;
    const int j = isB ? R - kLmax : R;
    if (R < 2 * kLmax && (FULL || j < L))
      *reinterpret_cast<i32x4 *>(tileA + (isB ? kTileBytes : 0) + (q >= CPP ? kLoOff : 0) + plane_off(j, q % CPP)) = t.v[i];
  }
}

__device__ __forceinline__ void lds_zero(char *p, int bytes, int lane) {
  int *z = reinterpret_cast<int *>(p);
  for (int i = lane; i < bytes / 4; i += AMPCONV_WAVE) z[i] = 0;
}

// channel-product fragment of row tile mt of ONE plane image.  Tile 1 uses the quarter map of mfma_tile.h: MFMA row m
// <-> token 16 + (m >> 2), so in C/D layout lane group g holds token 16 + g in reg 0
__device__ __forceinline__ i32x4 rowfrag(const char *img, int mt, int lane) {
  const int m = lane & 15, kg = lane >> 4;
  const int j = mt == 0 ? m : 16 + (m >> 2);
  return *reinterpret_cast<const i32x4 *>(img + plane_off(j, kg));
}
// ... of column tile nt straight from global memory (the unit's own side, once per unit; plain map: column m of tile 1
// is token 16 + m; token rows >= L read as zero).  `base` = the row-0 slot of the (node, head), `plane` = 0 / 64
template <bool HALF>
__device__ __forceinline__ i32x4 rowfrag_global(const char *base, unsigned row_bytes, int nt, int plane, int L, int lane) {
  const int m = lane & 15, kg = lane >> 4;
  const int j = nt == 0 ? m : 16 + m;
  i32x4 x = {0, 0, 0, 0};
  if (j < L && (!HALF || kg < 2)) x = *reinterpret_cast<const i32x4 *>(base + (unsigned)j * row_bytes + plane + 16 * kg);
  return x;
}
// token-product fragment of channel tile mc of ONE plane image: k-slots 0..3 = tokens 4 kg .. 4 kg + 3, slot 4 = token
// 16 + kg (slots 5..7 repeat it; the other operand holds zeros there)
__device__ __forceinline__ i32x4 colfrag(const char *img, int mc, int lane) {
  const int q = (lane >> 2) & 3, pp = lane & 3, kg = lane >> 4;
  const int ch = 2 * mc + (pp >> 1), half = (pp & 1) << 3;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4 *)(img + plane_off(4 * kg + q, ch) + half));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4 *)(img + plane_off(16 + kg, ch) + half));
  const i32x2 ai = __builtin_bit_cast(i32x2, a), bi = __builtin_bit_cast(i32x2, b);
  return i32x4{ai[0], ai[1], bi[0], bi[1]};
}
// every transposed fragment of a product group is in its registers before the group's first MFMA issues, and no
// transposed read is scheduled in among the MFMAs (the fence of proj_gemm.hip / edge_mfma_bf16.hip; tests/test_abi.py
// scans the shipped ISA for the pattern)
#define TR_FRAG_FENCE()                                    \
  do {                                                     \
    __builtin_amdgcn_sched_barrier(0);                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)

// behind a group of MFMAs whose fragment registers the next group's transposed reads may take over: the last MFMA has
// fetched its operands before a read redefines them (tools/scan_tr_hazard.py --gate, rule WAR)
#define TR_WAR_GUARD()                         \
  do {                                         \
    __builtin_amdgcn_sched_barrier(0);         \
    asm volatile("s_nop 7" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);         \
  } while (0)

// in front of a group of transposed reads that follows MFMAs of another phase (their fragment registers may be taken
// over): eight idle cycles and a compiler barrier for memory operations only -- the vector work around it still moves
#define TR_PRE_READ() asm volatile("s_nop 7" ::: "memory")

// C/D registers (already in the split's units, |x| < 2^16) -> the two planes of a token-product fragment: slots 0..3 =
// t0 (tokens 4 g + q), slot 4 = t1_0 (token 16 + g), slots 5..7 zero
__device__ __forceinline__ void cd_frag2(const f32x4 &t0, float t1_0, i32x4 &hi, i32x4 &lo) {
  int h0, l0, h1, l1, h2, l2;
  split2(t0[0], t0[1], h0, l0);
  split2(t0[2], t0[3], h1, l1);
  split2(t1_0, 0.f, h2, l2);
  hi = i32x4{h0, h1, h2, 0};
  lo = i32x4{l0, l1, l2, 0};
}

// softmax over the 20 source tokens of one destination-token column; `sc` = log2e / sqrt(dh) / (scale of Q' K'^T) is
// applied to the raw scores here, the result leaves multiplied by `mul`.  t0[q] = token 4 g + q, t1[0] = token 16 + g
// (regs 1..3 of tile 1 replicate it and come out as 0)
// Returns m * sc + log2(sum): P = exp2(S' * sc - that), what the source pass needs to rebuild P without reducing again.
template <bool FULL>
__device__ __forceinline__ float column_softmax(f32x4 &t0, f32x4 &t1, float sc, float mul, int L, int g) {
  if (!FULL) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (4 * g + q >= L) t0[q] = kMasked;
    if (16 + g >= L) t1[0] = kMasked;
  }
  float m = fmaxf(fmaxf(fmaxf(t0[0], t0[1]), fmaxf(t0[2], t0[3])), t1[0]);
  m = groups_max(m);
#pragma unroll
  for (int q = 0; q < 4; ++q) t0[q] = fast_exp2((t0[q] - m) * sc);
  t1[0] = fast_exp2((t1[0] - m) * sc);
  float l = ((t0[0] + t0[1]) + (t0[2] + t0[3])) + t1[0];
  l = groups_sum(l);
  const float inv = fast_rcp(l) * mul;
#pragma unroll
  for (int q = 0; q < 4; ++q) t0[q] *= inv;
  t1[0] *= inv;
  t1[1] = t1[2] = t1[3] = 0.f;
  return fmaf(m, sc, __builtin_amdgcn_logf(l));
}

struct Args {
  ampconv_view_t Q, K, V, dO, O, dK, dV;   // O = forward output / dQ (fp32); Q, K, V, dO: planes
  const int32_t *ptr, *idx;
  const float *bounds;                      // device: {bound of |Q|K|V|, bound of |dObar|: the planes' scales;
                                            //          recorded max |V|, recorded max |dObar|: the scale of dS}
  float *absmax;                            // or null: atomic max of the finite magnitudes written (backward passes)
  // softmax statistics handed from the destination pass to the source pass (kStatsPerUnit floats per (CSC position,
  // head): 20 log2-sum-exp values, then 20 delta values in the units of dP' = dO' V'^T), or null
  const int32_t *spos;                      // CSR position -> CSC position (destination pass)
  float *stats;
  HubArgs hub;
  int64_t n_units;
  int L, H;
};

__device__ __forceinline__ const char *slot_ptr(const ampconv_view_t &v, int64_t n, int h) {
  return reinterpret_cast<const char *>(v.ptr) + 4 * (n * v.node_stride + (int64_t)h * v.head_stride);
}

// output tile store (fp32): C/D layout lane (n = lane & 15, g), reg q -> channel 4 g + q + 16 mc, token of column n of
// tile nt: n / 16 + n (plain map) or, QUARTER (the source pass's own tokens sit on quarter-mapped columns), 16 + (n >> 2)
// for the lanes n % 4 == 0 of tile 1.  Returns the largest finite magnitude stored.
template <bool QUARTER, int MC>
__device__ __forceinline__ float store_tile(const ampconv_view_t &v, int64_t node, int h, const f32x4 (&T)[MC][2],
                                            float scale, int L, int lane) {
  const int g = lane >> 4, n = lane & 15;
  float m = 0.f;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int i = nt == 0 ? n : (QUARTER ? 16 + (n >> 2) : 16 + n);
    if (i < L && (nt == 0 || !QUARTER || (n & 3) == 0)) {
#pragma unroll
      for (int mc = 0; mc < MC; ++mc) {
        const float4 o = make_float4(T[mc][nt][0] * scale, T[mc][nt][1] * scale, T[mc][nt][2] * scale,
                                     T[mc][nt][3] * scale);
        const int64_t off = node * v.node_stride + (int64_t)h * v.head_stride + (int64_t)i * v.row_stride + 4 * g + 16 * mc;
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(v.ptr) + off) = o;
        m = finite_abs_max(m, o);
      }
    }
  }
  return m;
}

template <bool HALF>
__device__ __forceinline__ void store_zero_tile(const ampconv_view_t &v, int64_t node, int h, int L, int lane) {
  constexpr int MC = HALF ? 1 : 2;
  f32x4 Z[MC][2];
#pragma unroll
  for (int mc = 0; mc < MC; ++mc) Z[mc][0] = Z[mc][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  store_tile<false, MC>(v, node, h, Z, 0.f, L, lane);
}

// ---------------------------------------------------------------- forward
template <bool FULL, bool HALF>
__global__ __launch_bounds__(64 * kWavesPerBlock, 4) void fwd_f16x2(Args a) {
  constexpr int MC = HALF ? 1 : 2, kLo = HALF ? 32 : 64;          // channel tiles; byte offset of the lo plane in a slot
  constexpr float kIs = HALF ? 0.25f : 0.17677669529663687f;     // 1 / sqrt(dh)
  __shared__ __attribute__((aligned(16))) char lds_all[kWavesPerBlock][2 * kTileBytes];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.n_units) return;
  int64_t r, onode;
  int h, beg, end, deg;
  if (!map_unit(a.hub, a.ptr, unit, a.n_units, a.H, r, onode, h, beg, end, deg)) return;
  if (beg >= end && a.hub.mode != 2) return store_zero_tile<HALF>(a.O, onode, h, a.L, lane);
  const int L = a.L, g = lane >> 4;
  char *Kt = lds_all[wave], *Vt = Kt + kTileBytes;
  const float uq = 1.f / plane_scale(a.bounds[0]);           // exact: a power of two
  const float sc = (kLog2e * kIs * uq) * uq;   // log2e / sqrt(32) / (scale of Q' K'^T)

  i32x4 qh[2], ql[2];
  {
    const char *qb = slot_ptr(a.Q, r, h);
    const unsigned rb = (unsigned)a.Q.row_stride * 4u;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      qh[nt] = rowfrag_global<HALF>(qb, rb, nt, 0, L, lane);
      ql[nt] = rowfrag_global<HALF>(qb, rb, nt, kLo, L, lane);
    }
  }
  if (!FULL || HALF) lds_zero(Kt, 2 * kTileBytes, lane);
  f32x4 OT[MC][2];
#pragma unroll
  for (int mc = 0; mc < MC; ++mc) OT[mc][0] = OT[mc][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  PairRegsP kv;
  IdxWindow win;
  const unsigned krb = (unsigned)a.K.row_stride * 4u, vrb = (unsigned)a.V.row_stride * 4u;
  auto fetch = [&](int p) {
    const int64_t s = idxwin_get<false>(win, a.idx, nullptr, p, end, lane, nullptr);
    pair_load_p<FULL, HALF>(kv, slot_ptr(a.K, s, h), krb, slot_ptr(a.V, s, h), vrb, L, lane);
  };
  if (beg < end) {
    idxwin_load<false>(win, a.idx, nullptr, beg, end, lane);
    fetch(beg);
  }
  for (int p = beg; p < end; ++p) {
    pair_to_lds_p<FULL, HALF>(Kt, kv, L, lane);
    if (p + 1 < end) fetch(p + 1);
    __builtin_amdgcn_wave_barrier();

    f32x4 S[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const i32x4 kh = rowfrag(Kt, mt, lane), kl = rowfrag(Kt + kLoOff, mt, lane);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) S[mt][nt] = mfma3(kh, kl, qh[nt], ql[nt], f32x4{0.f, 0.f, 0.f, 0.f});
    }
    i32x4 ph[2], pl[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) column_softmax<FULL>(S[0][nt], S[1][nt], sc, kPScale, L, g);
    i32x4 vh[MC], vl[MC];
    TR_PRE_READ();
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) {
      vh[mc] = colfrag(Vt, mc, lane);
      vl[mc] = colfrag(Vt + kLoOff, mc, lane);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) cd_frag2(S[0][nt], S[1][nt][0], ph[nt], pl[nt]);      // (in the shadow of the reads)
    TR_FRAG_FENCE();
#pragma unroll
    for (int mc = 0; mc < MC; ++mc)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) OT[mc][nt] = mfma3(vh[mc], vl[mc], ph[nt], pl[nt], OT[mc][nt]);
    __builtin_amdgcn_wave_barrier();
  }
  const bool hubp = a.hub.mode == 2;
  store_tile<false, MC>(a.O, onode, h, OT, kPUnscale * uq * (hubp ? 1.f : (deg > 0 ? 1.f / (float)deg : 0.f)), L, lane);
}

// ---------------------------------------------------------------- backward, destination pass
template <bool FULL, bool STATS, bool HALF>
__global__ __launch_bounds__(64 * kWavesPerBlock, 3) void bwd_dst_f16x2(Args a) {
  constexpr int MC = HALF ? 1 : 2, kLo = HALF ? 32 : 64;
  constexpr float kIs = HALF ? 0.25f : 0.17677669529663687f;
  __shared__ __attribute__((aligned(16))) char lds_all[kWavesPerBlock][2 * kTileBytes];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.n_units) return;
  int64_t r, onode;
  int h, beg, end, deg;
  if (!map_unit(a.hub, a.ptr, unit, a.n_units, a.H, r, onode, h, beg, end, deg)) return;
  if (beg >= end && a.hub.mode != 2) return store_zero_tile<HALF>(a.O, onode, h, a.L, lane);
  const int L = a.L, g = lane >> 4;
  char *Kt = lds_all[wave], *Vt = Kt + kTileBytes;
  const float uq = 1.f / plane_scale(a.bounds[0]), ug = 1.f / plane_scale(a.bounds[1]);
  const float sc = (kLog2e * kIs * uq) * uq;

  i32x4 qh[2], ql[2], gh[2], gl[2];
  {
    const char *qb = slot_ptr(a.Q, r, h), *gb = slot_ptr(a.dO, r, h);
    const unsigned qrb = (unsigned)a.Q.row_stride * 4u, grb = (unsigned)a.dO.row_stride * 4u;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      qh[nt] = rowfrag_global<HALF>(qb, qrb, nt, 0, L, lane);
      ql[nt] = rowfrag_global<HALF>(qb, qrb, nt, kLo, L, lane);
      gh[nt] = rowfrag_global<HALF>(gb, grb, nt, 0, L, lane);
      gl[nt] = rowfrag_global<HALF>(gb, grb, nt, kLo, L, lane);
    }
  }
  // dS scale of this unit: its own dObar rows against the bound of any V row
  const float ss = ds_scale(tile_max_norm2(gh[0], gh[1]), a.bounds[2] * plane_scale(a.bounds[0]));
  if (!FULL || HALF) lds_zero(Kt, 2 * kTileBytes, lane);
  f32x4 dQT[MC][2];
#pragma unroll
  for (int mc = 0; mc < MC; ++mc) dQT[mc][0] = dQT[mc][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  PairRegsP kv;
  IdxWindow win;
  const unsigned krb = (unsigned)a.K.row_stride * 4u, vrb = (unsigned)a.V.row_stride * 4u;
  // (STATS: the window's weight slot carries the bits of spos[p], the CSC position of the edge)
  const float *sposf = reinterpret_cast<const float *>(a.spos);
  float cposf = 0.f, cposf_next = 0.f;
  auto fetch = [&](int p) {
    const int64_t s = idxwin_get<STATS>(win, a.idx, sposf, p, end, lane, &cposf_next);
    pair_load_p<FULL, HALF>(kv, slot_ptr(a.K, s, h), krb, slot_ptr(a.V, s, h), vrb, L, lane);
  };
  if (beg < end) {
    idxwin_load<STATS>(win, a.idx, sposf, beg, end, lane);
    fetch(beg);
  }
  for (int p = beg; p < end; ++p) {
    pair_to_lds_p<FULL, HALF>(Kt, kv, L, lane);
    cposf = cposf_next;
    if (p + 1 < end) fetch(p + 1);
    __builtin_amdgcn_wave_barrier();

    f32x4 S[2][2], dP[2][2];
    float lse[2], dlt[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const i32x4 kh = rowfrag(Kt, mt, lane), kl = rowfrag(Kt + kLoOff, mt, lane);
      const i32x4 vh = rowfrag(Vt, mt, lane), vl = rowfrag(Vt + kLoOff, mt, lane);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        S[mt][nt] = mfma3(kh, kl, qh[nt], ql[nt], f32x4{0.f, 0.f, 0.f, 0.f});
        dP[mt][nt] = mfma3(vh, vl, gh[nt], gl[nt], f32x4{0.f, 0.f, 0.f, 0.f});
      }
    }
    i32x4 sh[2], sl[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      lse[nt] = column_softmax<FULL>(S[0][nt], S[1][nt], sc, 1.f, L, g);        // P^T; tile-1 regs 1..3 come out 0
      float part = S[1][nt][0] * dP[1][nt][0];
#pragma unroll
      for (int q = 0; q < 4; ++q) part = fmaf(S[0][nt][q], dP[0][nt][q], part);
      const float delta = groups_sum(part);
      dlt[nt] = delta;
#pragma unroll
      for (int q = 0; q < 4; ++q) S[0][nt][q] *= (dP[0][nt][q] - delta) * ss;      // dS^T in the unit's split scale
      S[1][nt][0] *= (dP[1][nt][0] - delta) * ss;
    }
    i32x4 ch[MC], cl[MC];
    TR_PRE_READ();
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) {
      ch[mc] = colfrag(Kt, mc, lane);
      cl[mc] = colfrag(Kt + kLoOff, mc, lane);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) cd_frag2(S[0][nt], S[1][nt][0], sh[nt], sl[nt]);
    TR_FRAG_FENCE();
#pragma unroll
    for (int mc = 0; mc < MC; ++mc)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) dQT[mc][nt] = mfma3(ch[mc], cl[mc], sh[nt], sl[nt], dQT[mc][nt]);
    if (STATS) {
      // every lane group holds the statistics of its column: lanes 0..15 have tokens 0..15 (tile 0), lanes 16..19
      // (group 1, columns 0..3) tokens 16..19 (tile 1): two 80-byte stores per edge and head
      float *st = a.stats + ((int64_t)__builtin_bit_cast(int, cposf) * a.H + h) * kStatsPerUnit;
      if (lane < L) {
        st[lane] = lane < 16 ? lse[0] : lse[1];
        st[kLmax + lane] = lane < 16 ? dlt[0] : dlt[1];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // dQ = sum dS K / sqrt(dh): undo the split scale, the scales of dP' (Q|K|V and dObar) and of K'; the hub pass leaves
  // 1/sqrt(dh) to the combine pass
  const bool hubp = a.hub.mode == 2;
  const float m = store_tile<false, MC>(a.O, onode, h, dQT, (((1.f / ss) * uq) * ug) * uq * (hubp ? 1.f : kIs),
                                    L, lane);
  if (a.absmax) wave_record_absmax(a.absmax, m);
}

// ---------------------------------------------------------------- backward, source pass
// The unit's OWN K / V tiles live in two more LDS images and are re-read per edge (the registers they would occupy are
// the difference between two and three waves per SIMD).  Their tokens sit on the MFMA COLUMNS, tile 1 in the quarter
// map (column n <-> token 16 + (n >> 2): every tail token four times, no row of the 20-row image is read out of
// range); the row softmax masks the replicas, the store takes the lanes n % 4 == 0.
constexpr int kSrcLds = 4 * kTileBytes + 2 * kLmax * 4 + 32;      // four tile images + the edge's statistics (padded to 16)
template <bool FULL, bool STATS, bool HALF>
__global__ __launch_bounds__(64 * kWavesPerBlock, 3) void bwd_src_f16x2(Args a) {
  constexpr int MC = HALF ? 1 : 2;
  constexpr float kIs = HALF ? 0.25f : 0.17677669529663687f;
  __shared__ __attribute__((aligned(16))) char lds_all[kWavesPerBlock][kSrcLds];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.n_units) return;
  int64_t s, onode;
  int h, beg, end, deg;
  if (!map_unit(a.hub, a.ptr, unit, a.n_units, a.H, s, onode, h, beg, end, deg)) return;
  if (beg >= end && a.hub.mode != 2) {
    store_zero_tile<HALF>(a.dK, onode, h, a.L, lane);
    return store_zero_tile<HALF>(a.dV, onode, h, a.L, lane);
  }
  const int L = a.L, n = lane & 15;
  char *Qt = lds_all[wave], *Gt = Qt + kTileBytes, *Ko = Gt + kTileBytes, *Vo = Ko + kTileBytes;
  const float uq = 1.f / plane_scale(a.bounds[0]), ug = 1.f / plane_scale(a.bounds[1]);
  const float sc = (kLog2e * kIs * uq) * uq;

  if (!FULL || HALF) lds_zero(Qt, 4 * kTileBytes, lane);
  PairRegsP qg;
  pair_load_p<FULL, HALF>(qg, slot_ptr(a.K, s, h), (unsigned)a.K.row_stride * 4u, slot_ptr(a.V, s, h),
                    (unsigned)a.V.row_stride * 4u, L, lane);
  pair_to_lds_p<FULL, HALF>(Ko, qg, L, lane);
  __builtin_amdgcn_wave_barrier();
  // dS scale of this unit: its own V rows against the bound of any dObar row
  const float ss = ds_scale(tile_max_norm2(rowfrag(Vo, 0, lane), rowfrag(Vo, 1, lane)), a.bounds[3] * plane_scale(a.bounds[1]));
  f32x4 dKT[MC][2], dVT[MC][2];
#pragma unroll
  for (int mc = 0; mc < MC; ++mc)
    dKT[mc][0] = dKT[mc][1] = dVT[mc][0] = dVT[mc][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  IdxWindow win;
  const unsigned qrb = (unsigned)a.Q.row_stride * 4u, grb = (unsigned)a.dO.row_stride * 4u;
  float *stl = reinterpret_cast<float *>(Vo + kTileBytes);      // the edge's 40 statistics, staged for the row reads
  float sv = 0.f;                                               // lane l < 40: statistic l of the edge in flight
  auto fetch = [&](int p) {
    const int64_t d = idxwin_get<false>(win, a.idx, nullptr, p, end, lane, nullptr);
    pair_load_p<FULL, HALF>(qg, slot_ptr(a.Q, d, h), qrb, slot_ptr(a.dO, d, h), grb, L, lane);
    if (STATS && lane < kStatsPerUnit) sv = a.stats[((int64_t)p * a.H + h) * kStatsPerUnit + lane];
  };
  if (beg < end) {
    idxwin_load<false>(win, a.idx, nullptr, beg, end, lane);
    fetch(beg);
  }
  // own tokens on the columns: tile 0 column n = token n, tile 1 column n = token 16 + (n >> 2), one lane per token
  const bool v0 = FULL || n < L, v1 = (n & 3) == 0 && 16 + (n >> 2) < L;
  for (int p = beg; p < end; ++p) {
    pair_to_lds_p<FULL, HALF>(Qt, qg, L, lane);
    if (STATS && lane < kStatsPerUnit) stl[lane] = sv;
    if (p + 1 < end) fetch(p + 1);
    __builtin_amdgcn_wave_barrier();

    f32x4 S[2][2], dP[2][2];
    {
      i32x4 kh[2], kl[2], vh[2], vl[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        kh[nt] = rowfrag(Ko, nt, lane);
        kl[nt] = rowfrag(Ko + kLoOff, nt, lane);
        vh[nt] = rowfrag(Vo, nt, lane);
        vl[nt] = rowfrag(Vo + kLoOff, nt, lane);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const i32x4 ah = rowfrag(Qt, mt, lane), al = rowfrag(Qt + kLoOff, mt, lane);
        const i32x4 bh = rowfrag(Gt, mt, lane), bl = rowfrag(Gt + kLoOff, mt, lane);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          S[mt][nt] = mfma3(ah, al, kh[nt], kl[nt], f32x4{0.f, 0.f, 0.f, 0.f});
          dP[mt][nt] = mfma3(bh, bl, vh[nt], vl[nt], f32x4{0.f, 0.f, 0.f, 0.f});
        }
      }
    }
    // row softmax over the source tokens (columns across the 16 lanes of a DPP row); rows: tile 0 reg q = destination
    // token 4 g + q, tile 1 reg 0 = token 16 + g (quarter map).  After this block S holds P * 2^14 (for dV) and dP holds
    // dS in the unit's split scale.
    if (STATS) {
      // element-wise with the statistics of the destination pass: rows of tile 0 = tokens 4 g .. 4 g + 3, tile 1 reg 0 =
      // token 16 + g
      const int g = lane >> 4;
      const f32x4 l0 = *reinterpret_cast<const f32x4 *>(stl + 4 * g), d0 = *reinterpret_cast<const f32x4 *>(stl + kLmax + 4 * g);
      const float l1 = stl[16 + g], d1 = stl[kLmax + 16 + g];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int q = 0; q < (mt == 0 ? 4 : 1); ++q) {
          const float ls = mt == 0 ? l0[q] : l1, dl = mt == 0 ? d0[q] : d1;
          // (rows of tokens >= L have no statistics: their weights are zero, not exp2 of whatever the buffer holds)
          const bool vr = FULL || (mt == 0 ? 4 * g + q : 16 + g) < L;
          const float s0 = (v0 && vr) ? S[mt][0][q] : kMasked, s1 = (v1 && vr) ? S[mt][1][q] : kMasked;
          const float p0 = fast_exp2(fmaf(s0, sc, -ls)), p1 = fast_exp2(fmaf(s1, sc, -ls));
          S[mt][0][q] = p0 * kPScale;
          S[mt][1][q] = p1 * kPScale;
          dP[mt][0][q] = p0 * (dP[mt][0][q] - dl) * ss;
          dP[mt][1][q] = p1 * (dP[mt][1][q] - dl) * ss;
        }
      }
    } else {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int q = 0; q < (mt == 0 ? 4 : 1); ++q) {
        const float s0 = v0 ? S[mt][0][q] : kMasked, s1 = v1 ? S[mt][1][q] : kMasked;
        const float m = row16_max(fmaxf(s0, s1));
        float p0 = fast_exp2((s0 - m) * sc), p1 = fast_exp2((s1 - m) * sc);
        const float rinv = fast_rcp(row16_sum(p0 + p1));
        p0 *= rinv;
        p1 *= rinv;
        const float delta = row16_sum(fmaf(p0, dP[mt][0][q], p1 * dP[mt][1][q]));
        S[mt][0][q] = p0 * kPScale;
        S[mt][1][q] = p1 * kPScale;
        dP[mt][0][q] = p0 * (dP[mt][0][q] - delta) * ss;
        dP[mt][1][q] = p1 * (dP[mt][1][q] - delta) * ss;
      }
    }
    }
    {
      i32x4 ph[2], pl[2], gh[MC], gl[MC];
      TR_PRE_READ();
#pragma unroll
      for (int mc = 0; mc < MC; ++mc) {
        gh[mc] = colfrag(Gt, mc, lane);
        gl[mc] = colfrag(Gt + kLoOff, mc, lane);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) cd_frag2(S[0][nt], S[1][nt][0], ph[nt], pl[nt]);
      TR_FRAG_FENCE();
#pragma unroll
      for (int mc = 0; mc < MC; ++mc)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) dVT[mc][nt] = mfma3(gh[mc], gl[mc], ph[nt], pl[nt], dVT[mc][nt]);
      TR_WAR_GUARD();
    }
    {
      i32x4 sh[2], sl[2], qh[MC], ql[MC];
#pragma unroll
      for (int mc = 0; mc < MC; ++mc) {
        qh[mc] = colfrag(Qt, mc, lane);
        ql[mc] = colfrag(Qt + kLoOff, mc, lane);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) cd_frag2(dP[0][nt], dP[1][nt][0], sh[nt], sl[nt]);
      TR_FRAG_FENCE();
#pragma unroll
      for (int mc = 0; mc < MC; ++mc)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) dKT[mc][nt] = mfma3(qh[mc], ql[mc], sh[nt], sl[nt], dKT[mc][nt]);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // dK = sum dS^T Q / sqrt(dh), dV = sum P^T dObar (1/deg is inside dObar); the hub pass leaves 1/sqrt(dh) to the combine
  const bool hubp = a.hub.mode == 2;
  float m = store_tile<true, MC>(a.dK, onode, h, dKT, (((1.f / ss) * uq) * ug) * uq * (hubp ? 1.f : kIs), L, lane);
  m = fmaxf(m, store_tile<true, MC>(a.dV, onode, h, dVT, kPUnscale * ug, L, lane));
  if (a.absmax) wave_record_absmax(a.absmax, m);
}

typedef void (*EdgeKernel)(Args);
// kernels[2 * half + full]
int launch(Args &a, int L, bool half, EdgeKernel const (&kernels)[4], hipStream_t stream) {
  const int64_t blocks = (a.n_units + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > INT32_MAX) return AMPCONV_E_BADARG;
  const dim3 grid((unsigned)blocks), block(64 * kWavesPerBlock);
  kernels[2 * half + (L == kLmax)]<<<grid, block, 0, stream>>>(a);
  return ampconv_launch_status();
}
int launch_fwd(Args &a, int L, bool half, hipStream_t st) {
  static const EdgeKernel k[4] = {fwd_f16x2<false, false>, fwd_f16x2<true, false>, fwd_f16x2<false, true>, fwd_f16x2<true, true>};
  return launch(a, L, half, k, st);
}
int launch_dst(Args &a, int L, bool half, hipStream_t st) {
  static const EdgeKernel ks[4] = {bwd_dst_f16x2<false, true, false>, bwd_dst_f16x2<true, true, false>,
                                   bwd_dst_f16x2<false, true, true>, bwd_dst_f16x2<true, true, true>};
  static const EdgeKernel kn[4] = {bwd_dst_f16x2<false, false, false>, bwd_dst_f16x2<true, false, false>,
                                   bwd_dst_f16x2<false, false, true>, bwd_dst_f16x2<true, false, true>};
  return a.stats ? launch(a, L, half, ks, st) : launch(a, L, half, kn, st);
}
int launch_src(Args &a, int L, bool half, hipStream_t st) {
  static const EdgeKernel ks[4] = {bwd_src_f16x2<false, true, false>, bwd_src_f16x2<true, true, false>,
                                   bwd_src_f16x2<false, true, true>, bwd_src_f16x2<true, true, true>};
  static const EdgeKernel kn[4] = {bwd_src_f16x2<false, false, false>, bwd_src_f16x2<true, false, false>,
                                   bwd_src_f16x2<false, false, true>, bwd_src_f16x2<true, false, true>};
  return a.stats ? launch(a, L, half, ks, st) : launch(a, L, half, kn, st);
}

// planes: 16-byte aligned, the head's dh channels one 4 dh-byte slot, rows a whole number of 16-byte pieces apart
inline bool plane_view_ok(const ampconv_view_t &v, int dh) {
  return v.ptr && ((uintptr_t)v.ptr % 16 == 0) && v.head_stride == dh && (v.node_stride % 4 == 0) && (v.row_stride % 4 == 0);
}
inline bool f32_view_ok(const ampconv_view_t &v) {
  return v.ptr && ((uintptr_t)v.ptr % 16 == 0) && (v.head_stride % 4 == 0) && (v.node_stride % 4 == 0) && (v.row_stride % 4 == 0);
}
// partial-tile view of the hub workspace (edge_api.hip): chunk c, token l, channel cc at P[(c * L + l) * D + cc]
ampconv_view_t partial_view(void *ws, int64_t tile, int64_t n_chunks, int L, int D, int H) {
  return ampconv_view_t{(float *)ws + tile * n_chunks * L * D, (int64_t)L * D, (int64_t)D, (int64_t)(D / H)};
}
int check_shape(int64_t n, int L, int D, int H, const float *bounds) {
  if (L <= 0 || D <= 0 || H <= 0 || D % H != 0 || n < 0 || !bounds) return AMPCONV_E_BADARG;
  if (L > kLmax || (D / H != DH && D / H != DH / 2)) return AMPCONV_E_DTYPE;
  return AMPCONV_OK;
}

}  // namespace

extern "C" int ampconv_planes_supported(int L, int D, int H) {
  return L >= 1 && L <= kLmax && H > 0 && D % H == 0 && (D / H == DH || D / H == DH / 2);
}

extern "C" int ampconv_fwd_edge_planes(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, const int32_t *rowptr,
                                       const int32_t *col, int64_t n_rows, int L, int D, int H, ampconv_view_t O,
                                       const void *hub_plan, int64_t hub_chunks, void *hub_ws, const float *bounds,
                                       void *stream) {
  if (int rc = check_shape(n_rows, L, D, H, bounds)) return rc;
  if (n_rows == 0) return AMPCONV_OK;
  if (!plane_view_ok(Q, D / H) || !plane_view_ok(K, D / H) || !plane_view_ok(V, D / H) || !f32_view_ok(O) || !rowptr) return AMPCONV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  Args a{};
  a.Q = Q; a.K = K; a.V = V; a.O = O;
  a.ptr = rowptr; a.idx = col; a.bounds = bounds; a.L = L; a.H = H;
  if (hub_plan && hub_chunks > 0 && hub_ws) {          // long segments: main + hub + combine
    a.hub = HubArgs{(const int32_t *)hub_plan, 1};
    a.n_units = n_rows * H;
    if (int rc = launch_fwd(a, L, D / H == DH / 2, st)) return rc;
    const ampconv_view_t P = partial_view(hub_ws, 0, hub_chunks, L, D, H);
    a.O = P;
    a.hub.mode = 2;
    a.n_units = hub_chunks * H;
    if (int rc = launch_fwd(a, L, D / H == DH / 2, st)) return rc;
    return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)P.ptr, O, rowptr, L, D, H, 1.f, 0, st);
  }
  a.hub = HubArgs{nullptr, 0};
  a.n_units = n_rows * H;
  return launch_fwd(a, L, D / H == DH / 2, st);
}

extern "C" int ampconv_bwd_edge_dst_planes(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dObar,
                                           const int32_t *rowptr, const int32_t *col, int64_t n_rows, int L, int D, int H,
                                           ampconv_view_t dQ, const void *hub_plan, int64_t hub_chunks, void *hub_ws,
                                           const float *bounds, const int32_t *spos, float *stats, float *out_absmax,
                                           void *stream) {
  if (int rc = check_shape(n_rows, L, D, H, bounds)) return rc;
  if (stats && (!spos || (uintptr_t)stats % 16 != 0)) return AMPCONV_E_BADARG;
  if (n_rows == 0) return AMPCONV_OK;
  if (!plane_view_ok(Q, D / H) || !plane_view_ok(K, D / H) || !plane_view_ok(V, D / H) || !plane_view_ok(dObar, D / H) || !f32_view_ok(dQ) || !rowptr)
    return AMPCONV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  Args a{};
  a.Q = Q; a.K = K; a.V = V; a.dO = dObar; a.O = dQ;
  a.ptr = rowptr; a.idx = col; a.bounds = bounds; a.absmax = out_absmax; a.L = L; a.H = H;
  a.spos = spos; a.stats = stats;
  if (hub_plan && hub_chunks > 0 && hub_ws) {
    a.hub = HubArgs{(const int32_t *)hub_plan, 1};
    a.n_units = n_rows * H;
    if (int rc = launch_dst(a, L, D / H == DH / 2, st)) return rc;
    const ampconv_view_t P = partial_view(hub_ws, 0, hub_chunks, L, D, H);
    a.O = P;
    a.absmax = nullptr;                                // partial tiles: the combine pass records what it writes
    a.hub.mode = 2;
    a.n_units = hub_chunks * H;
    if (int rc = launch_dst(a, L, D / H == DH / 2, st)) return rc;
    return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)P.ptr, dQ, nullptr, L, D, H, 1.f / sqrtf((float)(D / H)),
                               0, st, out_absmax);
  }
  a.hub = HubArgs{nullptr, 0};
  a.n_units = n_rows * H;
  return launch_dst(a, L, D / H == DH / 2, st);
}

extern "C" int ampconv_bwd_edge_src_planes(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dObar,
                                           const int32_t *cscptr, const int32_t *crow, int64_t n_src, int L, int D, int H,
                                           ampconv_view_t dK, ampconv_view_t dV, const void *hub_plan, int64_t hub_chunks,
                                           void *hub_ws, const float *bounds, const float *stats, float *out_absmax,
                                           void *stream) {
  if (int rc = check_shape(n_src, L, D, H, bounds)) return rc;
  if (stats && (uintptr_t)stats % 16 != 0) return AMPCONV_E_BADARG;
  if (n_src == 0) return AMPCONV_OK;
  if (!plane_view_ok(Q, D / H) || !plane_view_ok(K, D / H) || !plane_view_ok(V, D / H) || !plane_view_ok(dObar, D / H) || !f32_view_ok(dK) ||
      !f32_view_ok(dV) || !cscptr)
    return AMPCONV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  Args a{};
  a.Q = Q; a.K = K; a.V = V; a.dO = dObar; a.dK = dK; a.dV = dV;
  a.ptr = cscptr; a.idx = crow; a.bounds = bounds; a.absmax = out_absmax; a.L = L; a.H = H;
  a.stats = const_cast<float *>(stats);
  if (hub_plan && hub_chunks > 0 && hub_ws) {
    a.hub = HubArgs{(const int32_t *)hub_plan, 1};
    a.n_units = n_src * H;
    if (int rc = launch_src(a, L, D / H == DH / 2, st)) return rc;
    const ampconv_view_t PK = partial_view(hub_ws, 0, hub_chunks, L, D, H), PV = partial_view(hub_ws, 1, hub_chunks, L, D, H);
    a.dK = PK;
    a.dV = PV;
    a.absmax = nullptr;
    a.hub.mode = 2;
    a.n_units = hub_chunks * H;
    if (int rc = launch_src(a, L, D / H == DH / 2, st)) return rc;
    if (int rc = ampconv_hub_combine(hub_plan, hub_chunks, (const float *)PK.ptr, dK, nullptr, L, D, H,
                                     1.f / sqrtf((float)(D / H)), 0, st, out_absmax))
      return rc;
    return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)PV.ptr, dV, nullptr, L, D, H, 1.f, 0, st, out_absmax);
  }
  a.hub = HubArgs{nullptr, 0};
  a.n_units = n_src * H;
  return launch_src(a, L, D / H == DH / 2, st);
}
