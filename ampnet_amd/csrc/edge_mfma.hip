// MFMA edge-phase kernels for gfx950: L <= 20 tokens per node, dh in {16, 32}, fp32.
//
// One wavefront owns one (row, head) unit for the whole kernel: no workgroup barrier, no
// atomics, every output tile is written once by its owner.  Per edge the wave streams the
// source node's K and V head tiles (20 x dh floats each, whole 128-B / 64-B lines) through
// registers into a private, XOR-swizzled LDS image and cuts MFMA operands from it
// (mfma_tile.h).  All products run on v_mfma_f32_16x16x4_f32, i.e. exact fp32.
//
//   forward   S^T = K Q^T  (tokens of the source on MFMA rows, destination tokens on columns)
//             P^T = softmax over rows (in-register + 2 cross-lane steps)
//             O^T += V^T P^T           (P^T C/D registers are the B operand as they stand; the
//                                       sum over the in-edges of the destination accumulates
//                                       in the same MFMA accumulators = PyG's mean numerator)
// Reference arithmetic replaced: torch functional.py:6578 (scale), :6589 (QK^T), :6590
// (softmax), :6594 (PV) per edge, and the mean aggregation of amp_conv.py:11.
#include <cstdlib>
#include "mfma_tile.h"

namespace {

constexpr float kLog2e = 1.4426950408889634f;
#ifndef AMPCONV_WPB
#define AMPCONV_WPB 4
#endif
constexpr int kWavesPerBlock = AMPCONV_WPB;
// prefetch depth (edges in flight per wave) of the three kernels' register rings
#ifndef AMPCONV_PF_FWD
#define AMPCONV_PF_FWD 1
#endif
#ifndef AMPCONV_PF_DST
#define AMPCONV_PF_DST 1
#endif
#ifndef AMPCONV_PF_SRC
#define AMPCONV_PF_SRC 1
#endif
// addressing of the 4x4x1 phase-2 reads (mfma_tile.h, nt_accumulate): physical channel halves (fewer address registers,
// 2-way LDS bank conflicts) or logical halves (conflict-free)
#ifndef AMPCONV_NT_PHYS_FWD
#define AMPCONV_NT_PHYS_FWD true     // keeps the forward pass at 128 registers = 4 waves per SIMD (1 spill instead of 7)
#endif
#ifndef AMPCONV_NT_PHYS_DST
#define AMPCONV_NT_PHYS_DST false
#endif
#ifndef AMPCONV_NT_PHYS_SRC
#define AMPCONV_NT_PHYS_SRC false
#endif
#ifndef AMPCONV_PF_DST_T4
#define AMPCONV_PF_DST_T4 1      // main tiles in flight per wave in bwd_dst_mfma_t4 (1 or 2; 2 needs AMPCONV_DST_WAVES=2)
#endif
#ifndef AMPCONV_DST_WAVES
#define AMPCONV_DST_WAVES 3      // waves per SIMD the destination pass is compiled for (168 registers; 2 = 256)
#endif
#ifndef AMPCONV_FWD_WAVES
#define AMPCONV_FWD_WAVES 4      // waves per SIMD the forward pass is compiled for (128 registers; 3 = 168): 7.5 vs 8.1 ms
#endif
#ifndef AMPCONV_PF_FWD_T4
#define AMPCONV_PF_FWD_T4 1      // main tiles in flight per wave in fwd_mfma_t4 (1 or 2; 2 needs AMPCONV_FWD_WAVES=3)
#endif
#ifndef AMPCONV_PF_SRC_T4
#define AMPCONV_PF_SRC_T4 1      // edges in flight per wave in bwd_src_mfma_t4 (1, 2 or 4); 2 and 4 need AMPCONV_SRC_WAVES=2
#endif
// dh = 16 (BASELINE config 3): 4 waves per SIMD with one edge in flight, or 3 with a ring of AMPCONV_PF_SRC16 edges whose
// statistics travel with the tiles
#ifndef AMPCONV_DST16_WAVES
#define AMPCONV_DST16_WAVES 4
#endif
#ifndef AMPCONV_PF_DST16
#define AMPCONV_PF_DST16 1
#endif
#ifndef AMPCONV_SRC16_WAVES
#define AMPCONV_SRC16_WAVES 3     // 6.12 vs 6.37 ms at 100 k / 1 M, D=128, H=8
#endif
#ifndef AMPCONV_PF_SRC16
#define AMPCONV_PF_SRC16 2
#endif
#ifndef AMPCONV_SRC_WAVES
#define AMPCONV_SRC_WAVES 3      // waves per SIMD the source pass is compiled for: 3 = 168 registers (own K / V operands
                                 // re-read from LDS, statistics half an edge ahead), 2 = 256 (everything in registers,
                                 // prefetch ring of AMPCONV_PF_SRC_T4 edges with their statistics): 115.8 vs 117.5 ms at cfg4
#endif
#ifdef AMPCONV_NO_SCHED_FENCE
#define SCHED_FENCE()
#else
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#ifdef AMPCONV_SETPRIO
#define PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define PRIO(x)
#endif

struct FwdArgs {
  ampconv_view_t Q, K, V, O;
  const int32_t *rowptr, *col, *qidx;
  HubArgs hub;
  int64_t n_units;   // n_rows * H (hub pass: chunks * H)
  int L, H;
  float qscale;      // log2(e) / sqrt(dh)
};

// softmax over the 20 source tokens of one destination-token column held as
// (t0[0..3] = tokens 4g..4g+3, t1[0] = token 16+g) across the 4 lane groups g.
// Returns m + log2(sum): P = exp2(S' - lse), what the source pass needs to rebuild P without
// reducing again (kStatsPerUnit below).
template <bool FULL>
__device__ __forceinline__ float column_softmax(f32x4 &t0, f32x4 &t1, int L, int g) {
  if (!FULL) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (4 * g + q >= L) t0[q] = kNegBig;
    if (16 + g >= L) t1[0] = kNegBig;
  }
  float m = fmaxf(fmaxf(fmaxf(t0[0], t0[1]), fmaxf(t0[2], t0[3])), t1[0]);
  m = groups_max(m);
#pragma unroll
  for (int q = 0; q < 4; ++q) t0[q] = fast_exp2(t0[q] - m);
  t1[0] = fast_exp2(t1[0] - m);
  float l = (t0[0] + t0[1]) + (t0[2] + t0[3]) + t1[0];
  l = groups_sum(l);
  const float inv = fast_rcp(l);
#pragma unroll
  for (int q = 0; q < 4; ++q) t0[q] *= inv;
  t1[0] *= inv;
  return m + __builtin_amdgcn_logf(l);
}

template <int DH, bool FULL, int PF>
__global__ __launch_bounds__(64 * kWavesPerBlock) void fwd_mfma(FwdArgs a) {
  using C = TileCfg<DH>;
  __shared__ __attribute__((aligned(16))) float lds_all[kWavesPerBlock][2][C::TILE_FLOATS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: unit, row, head and tile bases live in SGPRs
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.n_units) return;
  int64_t r, onode;
  int h, beg, end, deg;
  if (!map_unit(a.hub, a.rowptr, unit, a.n_units, a.H, r, onode, h, beg, end, deg)) return;
  const int L = a.L, g = lane >> 4;
  float *Kt = lds_all[wave][0], *Vt = lds_all[wave][1];
  const int64_t d = a.qidx ? a.qidx[r] : r;

  // fixed side: Q^T as the B operand (columns = destination tokens), pre-scaled so that
  // softmax(S) = exp2(S' - max S') / sum
  float qB[2][C::KK];
  {
    const float *qb = tile_ptr<const float>(a.Q, d, h);
    rowop_from_global<DH>(qB[0], qb, a.Q.row_stride, 0, true, a.qscale, L, lane);
    rowop_from_global<DH>(qB[1], qb, a.Q.row_stride, 1, true, a.qscale, L, lane);
  }
  if (!FULL) {          // token rows >= L of the images are read by the MFMAs: keep them finite
    tile_zero<DH>(Kt, lane);
    tile_zero<DH>(Vt, lane);
  }

  f32x4 OT[C::MC][2];
#pragma unroll
  for (int mc = 0; mc < C::MC; ++mc) OT[mc][0] = OT[mc][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // register ring: the tiles of the next PF edges are in flight while one edge computes
  PairRegs<DH> ring[PF];
  IdxWindow win;
  if (beg < end) idxwin_load<false>(win, a.col, nullptr, beg, end, lane);
  auto fetch = [&](PairRegs<DH> &buf, int p) {
    const int64_t s = idxwin_get<false>(win, a.col, nullptr, p, end, lane, nullptr);
    pair_load<DH, FULL>(buf, tile_ptr<const float>(a.K, s, h), a.K.row_stride,
                        tile_ptr<const float>(a.V, s, h), a.V.row_stride, L, lane);
  };
#pragma unroll
  for (int k = 0; k < PF; ++k)
    if (beg + k < end) fetch(ring[k], beg + k);
  auto step = [&](PairRegs<DH> &kv, int p) {
    pair_to_lds<DH, FULL>(Kt, kv, 1.f, 1.f, L, lane);
    if (p + PF < end) fetch(kv, p + PF);
    __builtin_amdgcn_wave_barrier();

    // S^T tiles [source-token tile mt][destination-token tile nt]
    PRIO(1);
    f32x4 S[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float kA[C::KK];
      rowop_from_lds<DH>(kA, Kt, mt, lane);
      S[mt][0] = S[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < C::KK; ++kk) {
        S[mt][0] = MFMA16(kA[kk], qB[0][kk], S[mt][0]);
        S[mt][1] = MFMA16(kA[kk], qB[1][kk], S[mt][1]);
      }
    }
    PRIO(0);
    column_softmax<FULL>(S[0][0], S[1][0], L, g);
    column_softmax<FULL>(S[0][1], S[1][1], L, g);
    PRIO(1);

    // O^T[channel tile mc][destination-token tile nt] += V^T P^T
#pragma unroll
    for (int mc = 0; mc < C::MC; ++mc) {
      float vA[5];
      colop_from_lds<DH>(vA, Vt, mc, lane);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int s = 0; s < 4; ++s) OT[mc][nt] = MFMA16(vA[s], S[0][nt][s], OT[mc][nt]);
        OT[mc][nt] = MFMA16(vA[4], S[1][nt][0], OT[mc][nt]);
      }
    }
    PRIO(0);
    __builtin_amdgcn_wave_barrier();
  };
  for (int p0 = beg; p0 < end; p0 += PF) {
#pragma unroll
    for (int k = 0; k < PF; ++k)
      if (p0 + k < end) step(ring[k], p0 + k);
  }

  // O^T C/D layout: lane (i' = lane & 15, g), reg q -> channel 4g + q + 16 mc, token i' + 16 nt
  // hub pass: unnormalised partial tile, the combine pass applies 1/deg
  const float inv = a.hub.mode == 2 ? 1.f : (deg > 0 ? 1.f / (float)deg : 0.f);
  float *ob = tile_ptr<float>(a.O, onode, h);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int i = (lane & 15) + 16 * nt;
    if (i < L) {
#pragma unroll
      for (int mc = 0; mc < C::MC; ++mc) {
        float4 o = make_float4(OT[mc][nt][0] * inv, OT[mc][nt][1] * inv, OT[mc][nt][2] * inv,
                               OT[mc][nt][3] * inv);
        *reinterpret_cast<float4 *>(ob + (int64_t)i * a.O.row_stride + 4 * g + 16 * mc) = o;
      }
    }
  }
}

#ifdef AMPCONV_STAMPS
// diagnostic build: per-phase s_memtime sums of bwd_src_mfma, written to a device buffer that
// nothing else reads (tools/stamp_bwd_src.py); never part of the product build
__device__ unsigned long long g_stamp_sums[8 * 4096];
#define STAMP_DECL unsigned long long st_last, st_now, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_now)::"memory"); \
  __builtin_amdgcn_sched_barrier(0); st_acc[i] += st_now - st_last; st_last = st_now; } while (0)
#define STAMP_FLUSH(unit) do { if ((unit) < 4096 && lane == 0) for (int i_ = 0; i_ < 8; ++i_) \
  g_stamp_sums[(unit) * 8 + i_] = st_acc[i_]; } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH(unit)
#endif

#ifdef AMPCONV_NO_ABSMAX      // A/B build without the out_absmax code (register pressure of the backward kernels)
constexpr bool kRecordAbsmax = false;
#else
constexpr bool kRecordAbsmax = true;
#endif
struct BwdArgs;
// BwdArgs::absmax, read from the kernel-argument segment where it is used (the kernel's epilogue) instead of being held
// in scalar registers through the edge loop.  Only the DESTINATION pass records: the source pass sits exactly at its 168
// registers (three waves per SIMD), the same code there spills two loop-invariant LDS addresses whose reloads carry a
// `vmcnt(0)` -- draining the prefetched tiles twice per edge, +4.5 % on the dominant kernel -- so the C entry point runs a
// pass over dK | dV instead (edge_api.hip)
__device__ __forceinline__ float *late_absmax_arg();
struct BwdArgs {
  ampconv_view_t Q, K, V, dO, dQ, dK, dV;
  HubArgs hub;
  const int32_t *ptr;      // rowptr (dst pass) / cscptr (src pass)
  const int32_t *idx;      // col (dst pass) / crow (src pass)
  const float *cinv;       // src pass: 1/in-degree of the destination of each CSC edge
  const int32_t *spos;     // dst pass, STATS: CSC position of each CSR edge
  float *stats;            // STATS: [CSC position][head][lse(20) | delta(20)], written by the dst pass
  int64_t n_units;
  int L, H;
  float qscale;            // log2(e) / sqrt(dh)
  float oscale;            // 1 / sqrt(dh) (dst pass) or ln 2 (src pass): final factor of dQ / dK
  float *absmax;           // dst pass: nullptr, or the running maximum of the finite magnitudes written to dQ
};

__device__ __forceinline__ float *late_absmax_arg() {
  const char __attribute__((address_space(4))) *ka =
      (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
  float *p;
  asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(p) : "s"(ka), "n"(offsetof(BwdArgs, absmax)));
  return p;
}

// ---- backward, destination pass: dQ[r], one wave per (destination, head).
//   S^T = K Q^T, P^T = softmax;  dP^T = V dO^T;  delta = colsum(P^T o dP^T);
//   dS^T = P^T o (dP^T - delta);  dQ^T += K^T dS^T   (dS^T C/D registers = B operand)
#ifdef AMPCONV_OCC
#define LB_DST __launch_bounds__(64 * kWavesPerBlock, 4)
#define LB_SRC __launch_bounds__(64 * kWavesPerBlock, 3)
#else
#define LB_DST __launch_bounds__(64 * kWavesPerBlock)
#define LB_SRC __launch_bounds__(64 * kWavesPerBlock)
#endif
template <int DH, bool FULL, int PF, bool STATS>
__global__ LB_DST void bwd_dst_mfma(BwdArgs a) {
  using C = TileCfg<DH>;
  __shared__ __attribute__((aligned(16))) float lds_all[kWavesPerBlock][2][C::TILE_FLOATS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: unit, row, head and tile bases live in SGPRs
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.n_units) return;
  int64_t r, onode;
  int h, beg, end, deg;
  if (!map_unit(a.hub, a.ptr, unit, a.n_units, a.H, r, onode, h, beg, end, deg)) return;
  const int L = a.L, g = lane >> 4;
  float *Kt = lds_all[wave][0], *Vt = lds_all[wave][1];
  const float inv = deg > 0 ? 1.f / (float)deg : 0.f;       // dO is the gradient of the MEAN
  const float oscale = a.hub.mode == 2 ? 1.f : a.oscale;   // hub pass: the combine pass scales

  float qB[2][C::KK], dOB[2][C::KK];
  {
    const float *qb = tile_ptr<const float>(a.Q, r, h);
    const float *gb = tile_ptr<const float>(a.dO, r, h);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      rowop_from_global<DH>(qB[nt], qb, a.Q.row_stride, nt, true, a.qscale, L, lane);
      rowop_from_global<DH>(dOB[nt], gb, a.dO.row_stride, nt, true, inv, L, lane);
    }
  }
  if (!FULL) {
    tile_zero<DH>(Kt, lane);
    tile_zero<DH>(Vt, lane);
  }
  f32x4 dQT[C::MC][2];
#pragma unroll
  for (int mc = 0; mc < C::MC; ++mc) dQT[mc][0] = dQT[mc][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  PairRegs<DH> ring[PF];
  float ring_pos[PF];      // STATS: CSC position of the edge (int bits; rides in the window's weight slot)
  IdxWindow win;
  const float *wts = reinterpret_cast<const float *>(a.spos);
  if (beg < end) idxwin_load<STATS>(win, a.idx, wts, beg, end, lane);
  auto fetch = [&](PairRegs<DH> &buf, float &pos, int p) {
    const int64_t s = idxwin_get<STATS>(win, a.idx, wts, p, end, lane, &pos);
    pair_load<DH, FULL>(buf, tile_ptr<const float>(a.K, s, h), a.K.row_stride,
                        tile_ptr<const float>(a.V, s, h), a.V.row_stride, L, lane);
  };
#pragma unroll
  for (int k = 0; k < PF; ++k)
    if (beg + k < end) fetch(ring[k], ring_pos[k], beg + k);
  auto step = [&](PairRegs<DH> &kv, float &pos, int p) {
    pair_to_lds<DH, FULL>(Kt, kv, 1.f, 1.f, L, lane);
    float *sb = nullptr;
    if (STATS) sb = a.stats + ((int64_t)__builtin_bit_cast(int, pos) * a.H + h) * kStatsPerUnit;
    if (p + PF < end) fetch(kv, pos, p + PF);
    __builtin_amdgcn_wave_barrier();

    // the destination-token columns of the two column tiles are independent (the softmax runs
    // along the source tokens = MFMA rows), so the tiles are processed one after the other
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      PRIO(1);
      f32x4 S0, S1, dP0, dP1;                    // row tiles 0 / 1 of this column tile
      S0 = S1 = dP0 = dP1 = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        float kA[C::KK], vA[C::KK];
        rowop_from_lds<DH>(kA, Kt, 0, lane);
        rowop_from_lds<DH>(vA, Vt, 0, lane);
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) {
          S0 = MFMA16(kA[kk], qB[nt][kk], S0);
          dP0 = MFMA16(vA[kk], dOB[nt][kk], dP0);
        }
        rowop_from_lds<DH>(kA, Kt, 1, lane);
        rowop_from_lds<DH>(vA, Vt, 1, lane);
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) {
          S1 = MFMA16(kA[kk], qB[nt][kk], S1);
          dP1 = MFMA16(vA[kk], dOB[nt][kk], dP1);
        }
      }
      PRIO(0);
      const float lse = column_softmax<FULL>(S0, S1, L, g);
      float part = S1[0] * dP1[0];
#pragma unroll
      for (int q = 0; q < 4; ++q) part = fmaf(S0[q], dP0[q], part);
      const float delta = groups_sum(part);
      if (STATS) {        // every lane group holds the column's values; group 0 stores them
        const int i = (lane & 15) + 16 * nt;
        if (g == 0 && i < kLmax) {
          sb[i] = lse;
          sb[kLmax + i] = delta;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) S0[q] *= dP0[q] - delta;     // S now holds dS^T
      S1[0] *= dP1[0] - delta;
      PRIO(1);
#pragma unroll
      for (int mc = 0; mc < C::MC; ++mc) {
        float kC[5];
        colop_from_lds<DH>(kC, Kt, mc, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) dQT[mc][nt] = MFMA16(kC[t], S0[t], dQT[mc][nt]);
        dQT[mc][nt] = MFMA16(kC[4], S1[0], dQT[mc][nt]);
      }
    }
    PRIO(0);
    __builtin_amdgcn_wave_barrier();
  };
  for (int p0 = beg; p0 < end; p0 += PF) {
#pragma unroll
    for (int k = 0; k < PF; ++k)
      if (p0 + k < end) step(ring[k], ring_pos[k], p0 + k);
  }

  float *ob = tile_ptr<float>(a.dQ, onode, h);
  float wmax = 0.f;          // largest finite magnitude this lane stores (recorded at the end if the caller asked)
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int i = (lane & 15) + 16 * nt;
    if (i < L) {
#pragma unroll
      for (int mc = 0; mc < C::MC; ++mc) {
        float4 o = make_float4(dQT[mc][nt][0] * oscale, dQT[mc][nt][1] * oscale,
                               dQT[mc][nt][2] * oscale, dQT[mc][nt][3] * oscale);
        *reinterpret_cast<float4 *>(ob + (int64_t)i * a.dQ.row_stride + 4 * g + 16 * mc) = o;
        if (kRecordAbsmax) wmax = finite_abs_max(wmax, o);
      }
    }
  }
  if (kRecordAbsmax) {
    float *const amax_p = late_absmax_arg();
    if (amax_p) wave_record_absmax(amax_p, wmax);
  }      // (partial-tile passes get no pointer: the combine pass records)
}

// ---- backward, source pass: dK[s], dV[s], one wave per (source, head) over the CSC.
//   S = Q K^T (destination tokens on MFMA rows, source tokens on columns), P = row softmax
//   (16-lane DPP reductions), dP = dO V^T, dS = P o (dP - delta);
//   dV^T += dO^T P,  dK^T += Q^T dS      (P / dS C/D registers = B operands)
struct StatRegs {      // lse / delta of the destination tokens this lane's C/D rows hold
  f32x4 l4, d4;        // tokens 4g .. 4g+3
  float l1, d1;        // token 16 + g
};

template <int DH, bool FULL, int PF, bool STATS>
__global__ LB_SRC void bwd_src_mfma(BwdArgs a) {
  using C = TileCfg<DH>;
  __shared__ __attribute__((aligned(16))) float lds_all[kWavesPerBlock][2][C::TILE_FLOATS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: unit, row, head and tile bases live in SGPRs
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.n_units) return;
  int64_t s, onode;
  int h, beg, end, deg;
  if (!map_unit(a.hub, a.ptr, unit, a.n_units, a.H, s, onode, h, beg, end, deg)) return;
  const int L = a.L, n = lane & 15;
  float *Qt = lds_all[wave][0], *Gt = lds_all[wave][1];
  const float oscale = a.hub.mode == 2 ? 1.f : a.oscale;

  float kB[2][C::KK], vB[2][C::KK];
  {
    const float *kb = tile_ptr<const float>(a.K, s, h);
    const float *vb = tile_ptr<const float>(a.V, s, h);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      rowop_from_global<DH>(kB[nt], kb, a.K.row_stride, nt, true, 1.f, L, lane);
      rowop_from_global<DH>(vB[nt], vb, a.V.row_stride, nt, true, 1.f, L, lane);
    }
  }
  if (!FULL) {
    tile_zero<DH>(Qt, lane);
    tile_zero<DH>(Gt, lane);
  }
  f32x4 dKT[C::MC][2], dVT[C::MC][2];
#pragma unroll
  for (int mc = 0; mc < C::MC; ++mc)
    dKT[mc][0] = dKT[mc][1] = dVT[mc][0] = dVT[mc][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  STAMP_DECL
  PairRegs<DH> ring[PF];
  float ring_inv[PF];
  StatRegs ring_st[PF];
  IdxWindow win;
  if (beg < end) idxwin_load<true>(win, a.idx, a.cinv, beg, end, lane);
  auto fetch = [&](PairRegs<DH> &buf, float &inv, StatRegs &st, int p) {
    const int64_t d = idxwin_get<true>(win, a.idx, a.cinv, p, end, lane, &inv);
    pair_load<DH, FULL>(buf, tile_ptr<const float>(a.Q, d, h), a.Q.row_stride,
                        tile_ptr<const float>(a.dO, d, h), a.dO.row_stride, L, lane);
    if (STATS) {
      const float *sb = a.stats + ((int64_t)p * a.H + h) * kStatsPerUnit;
      const int g = lane >> 4;
      st.l4 = *reinterpret_cast<const f32x4 *>(sb + 4 * g);
      st.d4 = *reinterpret_cast<const f32x4 *>(sb + kLmax + 4 * g);
      st.l1 = sb[16 + g];
      st.d1 = sb[kLmax + 16 + g];
    }
  };
#pragma unroll
  for (int k = 0; k < PF; ++k)
    if (beg + k < end) fetch(ring[k], ring_inv[k], ring_st[k], beg + k);
  auto step = [&](PairRegs<DH> &qg, float &inv, StatRegs &rs, int p) {
    STAMP(0);
    pair_to_lds<DH, FULL>(Qt, qg, a.qscale, inv, L, lane);
    STAMP(1);
    const StatRegs st = rs;
    if (p + PF < end) fetch(qg, inv, rs, p + PF);
    STAMP(2);
    __builtin_amdgcn_wave_barrier();

    PRIO(1);
    // the destination-token rows of the two row tiles are independent (the softmax runs along
    // the source tokens = lanes), so the tiles are processed one after the other: half the live
    // score registers
    const bool v0 = n < L, v1 = 16 + n < L;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x4 S0, S1, dP0, dP1;
      S0 = S1 = dP0 = dP1 = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        float qA[C::KK], gA[C::KK];
        rowop_from_lds<DH>(qA, Qt, mt, lane);
        rowop_from_lds<DH>(gA, Gt, mt, lane);
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) {
          S0 = MFMA16(qA[kk], kB[0][kk], S0);
          S1 = MFMA16(qA[kk], kB[1][kk], S1);
          dP0 = MFMA16(gA[kk], vB[0][kk], dP0);
          dP1 = MFMA16(gA[kk], vB[1][kk], dP1);
        }
      }
      STAMP(3);
      PRIO(0);
      // row softmax over the source tokens: columns n (+16 for tile 1) across the 16 lanes
#pragma unroll
      for (int q = 0; q < (mt == 0 ? 4 : 1); ++q) {
        if (STATS) {      // P and delta of this edge were reduced by the destination pass
          const float lse = mt == 0 ? st.l4[q] : st.l1, delta = mt == 0 ? st.d4[q] : st.d1;
          const float p0 = v0 ? fast_exp2(S0[q] - lse) : 0.f, p1 = v1 ? fast_exp2(S1[q] - lse) : 0.f;
          S0[q] = p0;
          S1[q] = p1;
          dP0[q] = p0 * (dP0[q] - delta);
          dP1[q] = p1 * (dP1[q] - delta);
          continue;
        }
        const float s0 = v0 ? S0[q] : kNegBig, s1 = v1 ? S1[q] : kNegBig;
        const float m = row16_max(fmaxf(s0, s1));
        float p0 = fast_exp2(s0 - m), p1 = fast_exp2(s1 - m);
        const float rinv = fast_rcp(row16_sum(p0 + p1));
        p0 *= rinv;
        p1 *= rinv;
        const float delta = row16_sum(fmaf(p0, dP0[q], p1 * dP1[q]));
        S0[q] = p0;
        S1[q] = p1;
        dP0[q] = p0 * (dP0[q] - delta);       // dP now holds dS
        dP1[q] = p1 * (dP1[q] - delta);
      }
      STAMP(4);
      PRIO(1);
      const int ks = lane >> 4;
#pragma unroll
      for (int t = (mt == 0 ? 0 : 4); t < (mt == 0 ? 4 : 5); ++t) {
        const int q = mt == 0 ? t : 0;
#pragma unroll
        for (int mc = 0; mc < C::MC; ++mc) {
          const int idx = lds_idx<DH>(col_token(t, ks), (lane & 15) + 16 * mc);
          const float gC = Gt[idx], qC = Qt[idx];
          dVT[mc][0] = MFMA16(gC, S0[q], dVT[mc][0]);
          dVT[mc][1] = MFMA16(gC, S1[q], dVT[mc][1]);
          dKT[mc][0] = MFMA16(qC, dP0[q], dKT[mc][0]);
          dKT[mc][1] = MFMA16(qC, dP1[q], dKT[mc][1]);
        }
      }
    }
    STAMP(5);
    PRIO(0);
    __builtin_amdgcn_wave_barrier();
  };
  for (int p0 = beg; p0 < end; p0 += PF) {
#pragma unroll
    for (int k = 0; k < PF; ++k)
      if (p0 + k < end) step(ring[k], ring_inv[k], ring_st[k], p0 + k);
  }

  STAMP_FLUSH(unit);
  const int g = lane >> 4;
  float *kb = tile_ptr<float>(a.dK, onode, h), *vb = tile_ptr<float>(a.dV, onode, h);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int j = n + 16 * nt;
    if (j < L) {
#pragma unroll
      for (int mc = 0; mc < C::MC; ++mc) {
        float4 k4 = make_float4(dKT[mc][nt][0] * oscale, dKT[mc][nt][1] * oscale,
                                dKT[mc][nt][2] * oscale, dKT[mc][nt][3] * oscale);
        float4 v4 = make_float4(dVT[mc][nt][0], dVT[mc][nt][1], dVT[mc][nt][2], dVT[mc][nt][3]);
        *reinterpret_cast<float4 *>(kb + (int64_t)j * a.dK.row_stride + 4 * g + 16 * mc) = k4;
        *reinterpret_cast<float4 *>(vb + (int64_t)j * a.dV.row_stride + 4 * g + 16 * mc) = v4;
      }
    }
  }
}

// ---- backward, source pass with the statistics of the destination pass, tail tokens batched.
// Tokens 16..19 of a destination fill only a quarter of a 16-row MFMA tile.  The source-side
// operands (K, V of this source) are the same for every edge and dK / dV are sums over the edges,
// so the quarter tiles of FOUR consecutive edges share one tile: MFMA row m <-> (edge m >> 2,
// token 16 + (m & 3)).  In C/D layout lane group g then holds edge g's four tail tokens in regs
// 0..3, and the products dV^T += dO^T P, dK^T += Q^T dS contract over (edge, token) jointly:
// k-slot g of step q takes its A operand from edge g's row 16 + q.  This needs no reduction
// along a row (P = exp2(S - lse), dS = P (dP - delta) are element-wise with the statistics), which
// is what makes it possible here and not in the passes that compute their own softmax.
// Per edge and head: 80 MFMAs instead of 104.
//
// LDS per wave: the two 16-row main images (rows 16..19 of the 20-row images stay unused) and a
// stash [4 edges][Q | dO][4 tokens][DH]; stash row rho = 8 e + 4 isG + t, chunk swizzle swz_tail.
// the stash swizzles like a 16-row image whose row 4 e + t is (edge e, token 16 + t): conflict-free for the row / column /
// 4x4x1 operand reads (round 1's function had 2-way conflicts on the first two) and the same chunk-bit-2 rule as the main
// images, which nt_accumulate relies on
template <int DH>
__device__ __forceinline__ int swz_tail(int rho) {
  return swz<DH>(4 * (rho >> 3) + (rho & 3));
}
template <int DH>
__device__ __forceinline__ int tail_idx(int rho, int c) {
  return rho * DH + ((((c >> 2) ^ swz_tail<DH>(rho)) << 2) | (c & 3));
}

// registers -> main images (token rows < 16) and the stash slot of edge e (rows 16..19), scaled
template <int DH, bool FULL, int IMG = TileCfg<DH>::TILE_FLOATS>
__device__ __forceinline__ void pair_to_lds_tail(float *ldsA, float *stash, int e, const PairRegs<DH> &t,
                                                 float mulA, float mulB, int L, int lane) {
  using C = TileCfg<DH>;
  const int r = lane / C::CH, q = lane % C::CH;
#pragma unroll
  for (int i = 0; i < PairRegs<DH>::NP; ++i) {
    const int R = r + C::RPI * i;
    const bool isB = R >= kLmax;
    const int j = isB ? R - kLmax : R;
    const bool valid = (R < 2 * kLmax) && (FULL || j < L);
    if (valid) {
      const float mul = isB ? mulB : mulA;
      float4 x = t.v[i];
      x.x *= mul; x.y *= mul; x.z *= mul; x.w *= mul;
      if (j < 16) {
        *reinterpret_cast<float4 *>(ldsA + (isB ? IMG : 0) + j * DH + ((q ^ swz<DH>(j)) << 2)) = x;
      } else {
        const int rho = 8 * e + (isB ? 4 : 0) + (j - 16);
        *reinterpret_cast<float4 *>(stash + rho * DH + ((q ^ swz_tail<DH>(rho)) << 2)) = x;
      }
    }
  }
}

// NT4: the source's own tail tokens 16..19 (the padded second COLUMN tile of S / dP and of the dK^T /
// dV^T accumulators, 25 % useful on 16x16x4) run on v_mfma_f32_4x4x1_16b_f32 instead (mfma_tile.h,
// "4-granular products"): per edge and head 40 16x16x4 + 40 4x4x1 instead of 80 16x16x4, i.e.
// ~1650 instead of 2560 matrix-pipe cycles.  The softmax is element-wise with the statistics, so the
// phase-1 result (one register: lane (g, sg, j) = destination token 4 sg + g, source token 16 + j)
// goes straight back in as the B operand of phase 2.
template <int DH, bool FULL, bool NT4>
__global__ __launch_bounds__(64 * kWavesPerBlock, DH == 32 ? AMPCONV_SRC_WAVES : AMPCONV_SRC16_WAVES) void bwd_src_mfma_t4(BwdArgs a) {
  using C = TileCfg<DH>;
  constexpr int kStash = 4 * 2 * 4 * DH;
  constexpr int NTM = NT4 ? 1 : 2;           // 16-wide column tiles on the 16x16x4 path
  // FIXED_LDS (the 168-register build, AMPCONV_SRC_WAVES=3): the source's own K / V tiles live in LDS and their
  // operands are re-read per phase (8 ds_read_b128) instead of holding 32 registers for the whole unit; the
  // streamed images then keep only the 16 rows the main phase reads (tokens 16..19 go to the stash anyway):
  // 13 312 B per wave = 3 blocks per CU
  constexpr bool FIXED_LDS = NT4 && DH == 32 && AMPCONV_SRC_WAVES >= 3;
  constexpr int kMain = FIXED_LDS ? 16 * DH : C::TILE_FLOATS;      // floats per streamed image
  constexpr int kFix = FIXED_LDS ? 2 * C::TILE_FLOATS : 0;
  __shared__ __attribute__((aligned(16))) float lds_all[kWavesPerBlock][2 * kMain + kStash + kFix];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.n_units) return;
  int64_t s, onode;
  int h, beg, end, deg;
  if (!map_unit(a.hub, a.ptr, unit, a.n_units, a.H, s, onode, h, beg, end, deg)) return;
  STAMP_DECL
  const int L = a.L, n = lane & 15, g = lane >> 4, sg = (lane >> 2) & 3, jt = lane & 3;
  float *Qt = lds_all[wave], *Gt = Qt + kMain, *stash = Qt + 2 * kMain;
  float *Kfix = stash + kStash, *Vfix = Kfix + C::TILE_FLOATS;
  const float oscale = a.hub.mode == 2 ? 1.f : a.oscale;

  float kB[FIXED_LDS ? 1 : NTM][FIXED_LDS ? 1 : C::KK], vB[FIXED_LDS ? 1 : NTM][FIXED_LDS ? 1 : C::KK];
  float kT[NT4 && !FIXED_LDS ? C::KK : 1], vT[NT4 && !FIXED_LDS ? C::KK : 1];
  {
    const float *kb = tile_ptr<const float>(a.K, s, h);
    const float *vb = tile_ptr<const float>(a.V, s, h);
    if constexpr (FIXED_LDS) {
      if (!FULL) {
        tile_zero<DH>(Kfix, lane);
        tile_zero<DH>(Vfix, lane);
      }
      TileRegs<DH> tk, tv;
      tile_load<DH>(tk, kb, a.K.row_stride, L, lane);
      tile_load<DH>(tv, vb, a.V.row_stride, L, lane);
      tile_to_lds<DH>(Kfix, tk, 1.f, L, lane);
      tile_to_lds<DH>(Vfix, tv, 1.f, L, lane);
    } else {
#pragma unroll
      for (int nt = 0; nt < NTM; ++nt) {
        rowop_from_global<DH>(kB[nt], kb, a.K.row_stride, nt, true, 1.f, L, lane);
        rowop_from_global<DH>(vB[nt], vb, a.V.row_stride, nt, true, 1.f, L, lane);
      }
      if constexpr (NT4) {
        tailop_from_global<DH>(kT, kb, a.K.row_stride, 1.f, L, lane);
        tailop_from_global<DH>(vT, vb, a.V.row_stride, 1.f, L, lane);
      }
    }
  }
  if (!FULL) {
    for (int i = lane; i < 2 * kMain; i += AMPCONV_WAVE) Qt[i] = 0.f;
  }
  // operands of the source's own tile for one phase: columns 0..15 (16x16x4 B operand) and 16..19 (4x4x1 B operand)
  auto fixed_ops = [&](float (&b0)[C::KK], float (&bt)[C::KK], const float *img, const float (&r0)[FIXED_LDS ? 1 : C::KK],
                       const float (&rt)[NT4 && !FIXED_LDS ? C::KK : 1]) {
    if constexpr (FIXED_LDS) {
      rowop_from_lds<DH>(b0, img, 0, lane);
      tailop_from_lds<DH>(bt, img, lane);
    } else {
#pragma unroll
      for (int kk = 0; kk < C::KK; ++kk) {
        b0[kk] = r0[kk];
        bt[kk] = rt[NT4 ? kk : 0];
      }
    }
  };
  // slots of a last, partial batch and token rows >= L are read by the MFMAs: keep them finite
  for (int i = lane; i < kStash; i += AMPCONV_WAVE) stash[i] = 0.f;
  f32x4 dKT[C::MC][NTM], dVT[C::MC][NTM];
  f32x4 dK4[C::MC], dV4[C::MC];              // NT4: block (g, sg), reg rr, lane jt: channel 16 hf + 4 g + rr, token 16 + jt
#pragma unroll
  for (int mc = 0; mc < C::MC; ++mc) {
#pragma unroll
    for (int nt = 0; nt < NTM; ++nt) dKT[mc][nt] = dVT[mc][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    dK4[mc] = dV4[mc] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // register ring: the tiles of the next PF edges are in flight while one edge computes (PF = 2 needs
  // the 256-register budget of two waves per SIMD: AMPCONV_SRC_WAVES=2)
  constexpr int PF = DH == 32 ? AMPCONV_PF_SRC_T4 : AMPCONV_PF_SRC16;
  PairRegs<DH> ring[PF];
  float ring_inv[PF];
  // statistics of an edge (lse / delta of its destination tokens 4g .. 4g+3 and, NT4, of token 4 sg + g).
  // STATS_AHEAD: they travel with the edge's tiles, PF edges ahead; otherwise they are loaded at the start
  // of the edge's own phase and the wave waits for them behind the first MFMAs (fewer live registers: what
  // the 168-register build needs; the wait is hidden only while other waves keep the matrix pipe busy)
  constexpr bool STATS_AHEAD = DH == 32 ? AMPCONV_SRC_WAVES < 3 : AMPCONV_SRC16_WAVES < 4;
  // STATS_HALF (the 168-register build): no ring slot to spare, so the statistics of edge p + 1 are requested right
  // after edge p's own have been consumed (end of phase 1) and arrive behind phase 2, ~60 % of an edge ahead of
  // their use, in the registers the element-wise step has just freed
  constexpr bool STATS_HALF = FIXED_LDS;
  struct EdgeStats { f32x4 l4, d4; float lT, dT; };
  EdgeStats ring_st[PF];
  IdxWindow win;
  auto load_stats = [&](EdgeStats &st, int p) {
    p = p < end ? p : end - 1;                 // STATS_HALF asks one edge past the segment
    const float *sbm = a.stats + ((int64_t)p * a.H + h) * kStatsPerUnit;
    st.l4 = *reinterpret_cast<const f32x4 *>(sbm + 4 * g);
    st.d4 = *reinterpret_cast<const f32x4 *>(sbm + kLmax + 4 * g);
    if constexpr (NT4) {
      st.lT = sbm[4 * sg + g];
      st.dT = sbm[kLmax + 4 * sg + g];
    }
  };
  auto fetch = [&](PairRegs<DH> &buf, float &inv, EdgeStats &st, int p) {
    const int64_t d = idxwin_get<true>(win, a.idx, a.cinv, p, end, lane, &inv);
    if constexpr (STATS_AHEAD) load_stats(st, p);
    pair_load<DH, FULL>(buf, tile_ptr<const float>(a.Q, d, h), a.Q.row_stride,
                        tile_ptr<const float>(a.dO, d, h), a.dO.row_stride, L, lane);
  };
  if (beg < end) {
    idxwin_load<true>(win, a.idx, a.cinv, beg, end, lane);
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      ring_inv[k] = 0.f;
      if (beg + k < end) fetch(ring[k], ring_inv[k], ring_st[k], beg + k);
    }
  }
  const bool v0 = n < L, v1 = 16 + n < L;
  const bool vt = FULL || 16 + jt < L;       // NT4: this lane's source tail token exists
  // NT4 phase 2: this lane's index of row 4 sg, channel 4 g + (lane & 3) in the main images / in the
  // stash (Q rows; the dO rows lie 4 rows further); the other rows and the second channel half
  // are XOR constants away (nt_accumulate)
  const int nt_main = lds_idx<DH>(4 * sg, 4 * g + jt), nt_stash = tail_idx<DH>(8 * sg, 4 * g + jt);

  // statistics of the batch's tail tile: tokens 16..19 of edge p0 + g (l4 / d4) and, NT4, token 16 + g of edge p0 + sg
  auto load_tail_stats = [&](EdgeStats &st, int p0) {
    const float *sb = a.stats + ((int64_t)(p0 + g < end ? p0 + g : end - 1) * a.H + h) * kStatsPerUnit;
    st.l4 = *reinterpret_cast<const f32x4 *>(sb + 16);
    st.d4 = *reinterpret_cast<const f32x4 *>(sb + kLmax + 16);
    if constexpr (NT4) {
      const float *sbT = a.stats + ((int64_t)(p0 + sg < end ? p0 + sg : end - 1) * a.H + h) * kStatsPerUnit;
      st.lT = sbT[16 + g];
      st.dT = sbT[kLmax + 16 + g];
    }
  };
  EdgeStats next_st;
  if constexpr (STATS_HALF) {
    if (beg < end) load_stats(next_st, beg);
  }
  for (int p0 = beg; p0 < end; p0 += 4) {
    EdgeStats tail_st;
    if constexpr (STATS_AHEAD) load_tail_stats(tail_st, p0);     // used four main phases later
#pragma unroll((PF == 1 && !FIXED_LDS) ? 1 : 4)
    for (int e = 0; e < 4; ++e) {
      const int p = p0 + e;
      if (p >= end) break;
      PairRegs<DH> &qg = ring[e % PF];             // batches are 4 edges long: the ring slot of an edge is e % PF
      float &inv_next = ring_inv[e % PF];
      STAMP(0);
      pair_to_lds_tail<DH, FULL, kMain>(Qt, stash, e, qg, a.qscale, inv_next, L, lane);
      STAMP(1);
      EdgeStats st = ring_st[e % PF];
      if constexpr (STATS_HALF) st = next_st;
      else if constexpr (!STATS_AHEAD) load_stats(st, p);   // ahead of the tile loads of the next edge
      const f32x4 l4 = st.l4, d4 = st.d4;
      const float lT = st.lT, dT = st.dT;
      if (p + PF < end) fetch(qg, inv_next, ring_st[e % PF], p + PF);
      __builtin_amdgcn_wave_barrier();
      STAMP(2);

      // main tile: destination tokens 0..15 of this edge.  S first, then dP (NT4: one after the other
      // keeps only one row operand and one pair of result tiles live at a time)
      f32x4 S0, S1, dP0, dP1;
      S0 = S1 = dP0 = dP1 = f32x4{0.f, 0.f, 0.f, 0.f};
      float pT = 0.f, dsT = 0.f;
      if constexpr (NT4) {
        {
          float qA[C::KK], b0[C::KK], bt[C::KK];
          rowop_from_lds<DH>(qA, Qt, 0, lane);
          fixed_ops(b0, bt, Kfix, kB[0], kT);
#pragma unroll
          for (int kk = 0; kk < C::KK; ++kk) {
            S0 = MFMA16(qA[kk], b0[kk], S0);
            S1 = MFMA4(qA[kk], bt[kk], S1, 0);          // 4x4x1 partials of the columns 16..19
          }
        }
        pT = (vt && (FULL || 4 * sg + g < L)) ? fast_exp2(reduce_transpose(S1) - lT) : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) S0[q] = v0 ? fast_exp2(S0[q] - l4[q]) : 0.f;
        SCHED_FENCE();
        STAMP(3);
        {
          float gA[C::KK], b0[C::KK], bt[C::KK];
          rowop_from_lds<DH>(gA, Gt, 0, lane);
          fixed_ops(b0, bt, Vfix, vB[0], vT);
#pragma unroll
          for (int kk = 0; kk < C::KK; ++kk) {
            dP0 = MFMA16(gA[kk], b0[kk], dP0);
            dP1 = MFMA4(gA[kk], bt[kk], dP1, 0);
          }
        }
        dsT = pT * (reduce_transpose(dP1) - dT);
#pragma unroll
        for (int q = 0; q < 4; ++q) dP0[q] = S0[q] * (dP0[q] - d4[q]);      // dP now holds dS
        if constexpr (STATS_HALF) {
          load_stats(next_st, p + 1);
        }
        STAMP(4);
      } else {
        {
          float qA[C::KK], gA[C::KK];
          rowop_from_lds<DH>(qA, Qt, 0, lane);
          rowop_from_lds<DH>(gA, Gt, 0, lane);
#pragma unroll
          for (int kk = 0; kk < C::KK; ++kk) {
            S0 = MFMA16(qA[kk], kB[0][kk], S0);
            S1 = MFMA16(qA[kk], kB[NTM - 1][kk], S1);
            dP0 = MFMA16(gA[kk], vB[0][kk], dP0);
            dP1 = MFMA16(gA[kk], vB[NTM - 1][kk], dP1);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float pa = v0 ? fast_exp2(S0[q] - l4[q]) : 0.f, pb = v1 ? fast_exp2(S1[q] - l4[q]) : 0.f;
          S0[q] = pa;
          S1[q] = pb;
          dP0[q] = pa * (dP0[q] - d4[q]);          // dP now holds dS
          dP1[q] = pb * (dP1[q] - d4[q]);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int mc = 0; mc < C::MC; ++mc) {
          const int idx = lds_idx<DH>(col_token(t, g), n + 16 * mc);
          const float gC = Gt[idx], qC = Qt[idx];
          dVT[mc][0] = MFMA16(gC, S0[t], dVT[mc][0]);
          dKT[mc][0] = MFMA16(qC, dP0[t], dKT[mc][0]);
          if constexpr (!NT4) {
            dVT[mc][NTM - 1] = MFMA16(gC, S1[t], dVT[mc][NTM - 1]);
            dKT[mc][NTM - 1] = MFMA16(qC, dP1[t], dKT[mc][NTM - 1]);
          }
        }
      }
      if constexpr (NT4) {
        nt_accumulate<DH, AMPCONV_NT_PHYS_SRC>(dV4, Gt, nt_main, pT);
        nt_accumulate<DH, AMPCONV_NT_PHYS_SRC>(dK4, Qt, nt_main, dsT);
      }
      __builtin_amdgcn_wave_barrier();
      STAMP(5);
    }

    // tail tile of the (up to) four edges p0 .. p0+3: lane group g <-> edge p0 + g
    {
      const bool live = p0 + g < end;
      // NT4: lane (g, sg, jt) <-> row 4 sg + g of the tail tile = (edge p0 + sg, destination token 16 + g)
      const bool liveT = p0 + sg < end;
      EdgeStats tl = tail_st;
      if constexpr (!STATS_AHEAD) load_tail_stats(tl, p0);
      const f32x4 l4 = tl.l4, d4 = tl.d4;
      const float lT = tl.lT, dT = tl.dT;
      f32x4 S0, S1, dP0, dP1;
      S0 = S1 = dP0 = dP1 = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        // ROW operand: MFMA row m = lane & 15 <-> stash row of (edge m >> 2, token 16 + (m & 3))
        const int rq = 8 * (n >> 2) + (n & 3), rg = rq + 4;
        float qA[C::KK], gA[C::KK];
#pragma unroll
        for (int b = 0; b < C::KK / 4; ++b) {
          const int chunk = (C::KK / 4) * g + b;
          const float4 x = *reinterpret_cast<const float4 *>(stash + rq * DH + ((chunk ^ swz_tail<DH>(rq)) << 2));
          const float4 y = *reinterpret_cast<const float4 *>(stash + rg * DH + ((chunk ^ swz_tail<DH>(rg)) << 2));
          qA[4 * b + 0] = x.x; qA[4 * b + 1] = x.y; qA[4 * b + 2] = x.z; qA[4 * b + 3] = x.w;
          gA[4 * b + 0] = y.x; gA[4 * b + 1] = y.y; gA[4 * b + 2] = y.z; gA[4 * b + 3] = y.w;
        }
        if constexpr (NT4) {
          float b0[C::KK], bt[C::KK];
          fixed_ops(b0, bt, Kfix, kB[0], kT);
#pragma unroll
          for (int kk = 0; kk < C::KK; ++kk) {
            S0 = MFMA16(qA[kk], b0[kk], S0);
            S1 = MFMA4(qA[kk], bt[kk], S1, 0);
          }
          fixed_ops(b0, bt, Vfix, vB[0], vT);
#pragma unroll
          for (int kk = 0; kk < C::KK; ++kk) {
            dP0 = MFMA16(gA[kk], b0[kk], dP0);
            dP1 = MFMA4(gA[kk], bt[kk], dP1, 0);
          }
        } else {
#pragma unroll
          for (int kk = 0; kk < C::KK; ++kk) {
            S0 = MFMA16(qA[kk], kB[0][kk], S0);
            dP0 = MFMA16(gA[kk], vB[0][kk], dP0);
            S1 = MFMA16(qA[kk], kB[NTM - 1][kk], S1);
            dP1 = MFMA16(gA[kk], vB[NTM - 1][kk], dP1);
          }
        }
      }
      float pT = 0.f, dsT = 0.f;
      if constexpr (NT4) {
        const float zs = reduce_transpose(S1), zd = reduce_transpose(dP1);
        pT = (vt && liveT && (FULL || 16 + g < L)) ? fast_exp2(zs - lT) : 0.f;
        dsT = pT * (zd - dT);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {       // C/D reg q of lane group g = token 16 + q of edge p0 + g
        const float pa = (v0 && live) ? fast_exp2(S0[q] - l4[q]) : 0.f;
        S0[q] = pa;
        dP0[q] = pa * (dP0[q] - d4[q]);
        if constexpr (!NT4) {
          const float pb = (v1 && live) ? fast_exp2(S1[q] - l4[q]) : 0.f;
          S1[q] = pb;
          dP1[q] = pb * (dP1[q] - d4[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {       // k-slot g of step q: row 16 + q of edge g's tiles
#pragma unroll
        for (int mc = 0; mc < C::MC; ++mc) {
          const float qC = stash[tail_idx<DH>(8 * g + q, n + 16 * mc)];
          const float gC = stash[tail_idx<DH>(8 * g + 4 + q, n + 16 * mc)];
          dVT[mc][0] = MFMA16(gC, S0[q], dVT[mc][0]);
          dKT[mc][0] = MFMA16(qC, dP0[q], dKT[mc][0]);
          if constexpr (!NT4) {
            dVT[mc][NTM - 1] = MFMA16(gC, S1[q], dVT[mc][NTM - 1]);
            dKT[mc][NTM - 1] = MFMA16(qC, dP1[q], dKT[mc][NTM - 1]);
          }
        }
      }
      if constexpr (NT4) {      // tail-tile row 4 sg + x = (edge sg, token 16 + x): stash row 8 sg + 4 isG + x
        nt_accumulate<DH, AMPCONV_NT_PHYS_SRC>(dV4, stash + 4 * DH, nt_stash, pT);
        nt_accumulate<DH, AMPCONV_NT_PHYS_SRC>(dK4, stash, nt_stash, dsT);
      }
      __builtin_amdgcn_wave_barrier();
      STAMP(6);
    }
  }
  STAMP(7);

  float *kb = tile_ptr<float>(a.dK, onode, h), *vb = tile_ptr<float>(a.dV, onode, h);
#pragma unroll
  for (int nt = 0; nt < NTM; ++nt) {
    const int j = n + 16 * nt;
    if (j < L) {
#pragma unroll
      for (int mc = 0; mc < C::MC; ++mc) {
        float4 k4 = make_float4(dKT[mc][nt][0] * oscale, dKT[mc][nt][1] * oscale,
                                dKT[mc][nt][2] * oscale, dKT[mc][nt][3] * oscale);
        float4 v4 = make_float4(dVT[mc][nt][0], dVT[mc][nt][1], dVT[mc][nt][2], dVT[mc][nt][3]);
        *reinterpret_cast<float4 *>(kb + (int64_t)j * a.dK.row_stride + 4 * g + 16 * mc) = k4;
        *reinterpret_cast<float4 *>(vb + (int64_t)j * a.dV.row_stride + 4 * g + 16 * mc) = v4;
      }
    }
  }
  if constexpr (NT4) {          // the four sg partial sums of every block column, then one quad stores
    nt_fix_halves<DH, AMPCONV_NT_PHYS_SRC>(dK4, lane);
    nt_fix_halves<DH, AMPCONV_NT_PHYS_SRC>(dV4, lane);
#pragma unroll
    for (int hf = 0; hf < C::MC; ++hf) {
      float4 k4, v4;
      k4.x = quads_sum(dK4[hf][0]) * oscale; k4.y = quads_sum(dK4[hf][1]) * oscale;
      k4.z = quads_sum(dK4[hf][2]) * oscale; k4.w = quads_sum(dK4[hf][3]) * oscale;
      v4.x = quads_sum(dV4[hf][0]); v4.y = quads_sum(dV4[hf][1]);
      v4.z = quads_sum(dV4[hf][2]); v4.w = quads_sum(dV4[hf][3]);
      if (sg == 0 && 16 + jt < L) {
        *reinterpret_cast<float4 *>(kb + (int64_t)(16 + jt) * a.dK.row_stride + 16 * hf + 4 * g) = k4;
        *reinterpret_cast<float4 *>(vb + (int64_t)(16 + jt) * a.dV.row_stride + 16 * hf + 4 * g) = v4;
      }
    }
  }
  STAMP_FLUSH(unit);
}

// ---- forward with the tail tokens of four consecutive edges in one MFMA tile.
// The column softmax of an edge runs over its 20 source tokens, 16 in the main tile and 4 in the
// tail, so the tail scores must exist BEFORE the edge's main tile is normalised: per batch of four
// edges the tail tile S^T[(edge g, token 16+q)][dst token] comes first (its A operand = the four
// edges' K rows 16..19 as one 16-row LDS image, row rho = 4 edge + token - 16), each edge's main
// phase then folds "its" lane group of the tail tile into the same cross-group max / sum, and
// the batch ends with O^T += V_tail^T P_tail^T, contracting over (edge, token) jointly (V rows
// 16..19 as a second 16-row image).  The tail rows are fetched as whole rows (coalesced), the K
// rows a batch ahead; gathering them lane-by-lane straight into operand registers measured 46 %
// SLOWER than no batching (64 sector requests per instruction).  52 -> 40 MFMAs per edge and head.
template <int DH>
struct TailRegs {                      // rows 16..19 of one tensor for four edges: 16 rows
  static constexpr int NP = 16 / TileCfg<DH>::RPI;
  float4 v[NP];
};
template <int DH>
struct MainRegs {                      // rows 0..15 of two tiles: 32 rows
  static constexpr int NP = 32 / TileCfg<DH>::RPI;
  float4 v[NP];
};
template <int DH, bool FULL>
__device__ __forceinline__ void main_load(MainRegs<DH> &t, const float *baseA, int64_t strideA,
                                          const float *baseB, int64_t strideB, int L, int lane) {
  using C = TileCfg<DH>;
  const int r = lane / C::CH, q = lane % C::CH;
#pragma unroll
  for (int i = 0; i < MainRegs<DH>::NP; ++i) {
    const int R = r + C::RPI * i;
    const bool isB = R >= 16;
    const int j = R & 15;
    const unsigned boff = ((unsigned)j * (unsigned)(isB ? strideB : strideA) + 4u * (unsigned)q) * 4u;
    const char *p = reinterpret_cast<const char *>(isB ? baseB : baseA) + boff;
    if (FULL || j < L) t.v[i] = STREAM_LOAD4(p);
  }
}
template <int DH, bool FULL>
__device__ __forceinline__ void main_to_lds(float *ldsA, const MainRegs<DH> &t, int L, int lane) {
  using C = TileCfg<DH>;
  const int r = lane / C::CH, q = lane % C::CH;
#pragma unroll
  for (int i = 0; i < MainRegs<DH>::NP; ++i) {
    const int R = r + C::RPI * i;
    const bool isB = R >= 16;
    const int j = R & 15;
    if (FULL || j < L) {
      const float4 x = t.v[i];     // member-wise: an aggregate copy to LDS keeps the registers' struct in scratch
      *reinterpret_cast<float4 *>(ldsA + (isB ? 16 * DH : 0) + j * DH + ((q ^ swz<DH>(j)) << 2)) =
          make_float4(x.x, x.y, x.z, x.w);
    }
  }
}

// rows 16..19 of four edges (nodes id0..id3), whole rows: lane (r, q) owns 16 B of rows rho = r + RPI i
template <int DH, bool FULL>
__device__ __forceinline__ void tail_load(TailRegs<DH> &t, const ampconv_view_t &view, int h, int id0, int id1,
                                          int id2, int id3, int L, int lane) {
  using C = TileCfg<DH>;
  const int r0 = lane / C::CH, q = lane % C::CH;
#pragma unroll
  for (int i = 0; i < TailRegs<DH>::NP; ++i) {
    const int rho = r0 + C::RPI * i, tok = 16 + (rho & 3);
    const int lo = (rho & 4) ? id1 : id0, hi = (rho & 4) ? id3 : id2;     // edge rho >> 2
    const int node = (rho & 8) ? hi : lo;
    if (FULL || tok < L)
      t.v[i] = STREAM_LOAD4(tile_ptr<const float>(view, node, h) + (int64_t)tok * view.row_stride + 4 * q);
  }
}
template <int DH, bool FULL>
__device__ __forceinline__ void tail_to_lds(float *img, const TailRegs<DH> &t, int L, int lane) {
  using C = TileCfg<DH>;
  const int r0 = lane / C::CH, q = lane % C::CH;
#pragma unroll
  for (int i = 0; i < TailRegs<DH>::NP; ++i) {
    const int rho = r0 + C::RPI * i;
    if (FULL || 16 + (rho & 3) < L) {
      const float4 x = t.v[i];
      *reinterpret_cast<float4 *>(img + rho * DH + ((q ^ swz<DH>(rho)) << 2)) = make_float4(x.x, x.y, x.z, x.w);
    }
  }
}

// NT4: the destination's own tail tokens 16..19 (the padded second COLUMN tile) run on
// v_mfma_f32_4x4x1_16b_f32 (mfma_tile.h, "4-granular products"): phase 1 leaves ONE register per 16-row
// tile, lane (g, sg, j) = S^T[source token 4 sg + g][destination token 16 + j]; the column softmax of edge e
// runs over the 16 lanes that share j (quad rotation + row swaps) with lane quad sg == e folding in the
// batch's tail tile (its row 4 sg + g = source token 16 + g of edge sg); phase 2 takes the probabilities
// back as the B operand.  Per edge and head: 20 16x16x4 + 20 4x4x1 instead of 40 16x16x4.
template <int DH, bool FULL, bool NT4>
__global__ __launch_bounds__(64 * kWavesPerBlock, AMPCONV_FWD_WAVES) void fwd_mfma_t4(FwdArgs a) {
  using C = TileCfg<DH>;
  constexpr int NTM = NT4 ? 1 : 2;           // 16-wide destination-token column tiles on the 16x16x4 path
  __shared__ __attribute__((aligned(16))) float lds_all[kWavesPerBlock][4 * 16 * DH];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.n_units) return;
  int64_t r, onode;
  int h, beg, end, deg;
  if (!map_unit(a.hub, a.rowptr, unit, a.n_units, a.H, r, onode, h, beg, end, deg)) return;
  const int L = a.L, g = lane >> 4, n = lane & 15, sg = (lane >> 2) & 3, jt = lane & 3;
  float *Kt = lds_all[wave], *Vt = Kt + 16 * DH, *Ktail = Kt + 32 * DH, *Vtail = Kt + 48 * DH;
  const int64_t d = a.qidx ? a.qidx[r] : r;

  float qB[NTM][C::KK];
  float qT[NT4 ? C::KK : 1];
  {
    const float *qb = tile_ptr<const float>(a.Q, d, h);
#pragma unroll
    for (int nt = 0; nt < NTM; ++nt) rowop_from_global<DH>(qB[nt], qb, a.Q.row_stride, nt, true, a.qscale, L, lane);
    if constexpr (NT4) tailop_from_global<DH>(qT, qb, a.Q.row_stride, a.qscale, L, lane);
  }
  if (!FULL)
    for (int i = lane; i < 4 * 16 * DH; i += AMPCONV_WAVE) Kt[i] = 0.f;
  f32x4 OT[C::MC][NTM];
  f32x4 O4[C::MC];                           // NT4: block (g, sg), reg rr, lane jt: channel 16 hf + 4 g + rr, token 16 + jt
#pragma unroll
  for (int mc = 0; mc < C::MC; ++mc) {
#pragma unroll
    for (int nt = 0; nt < NTM; ++nt) OT[mc][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    O4[mc] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int nt_base = lds_idx<DH>(4 * sg, 4 * g + jt);        // NT4 phase 2: row 4 sg, channel 4 g + jt of a 16-row image

  // sources of the edges of the current / next batch (wave-uniform scalars; positions clamped to
  // the segment, the window is walked strictly forwards)
  IdxWindow win;
  int id0 = 0, id1 = 0, id2 = 0, id3 = 0, nid0 = 0, nid1 = 0, nid2 = 0, nid3 = 0;
#define AMPCONV_IDS(o0, o1, o2, o3, p)                                                               \
  do {                                                                                               \
    const int p_ = (p);                                                                              \
    o0 = idxwin_get<false>(win, a.col, nullptr, p_ < end ? p_ : end - 1, end, lane, nullptr);         \
    o1 = idxwin_get<false>(win, a.col, nullptr, p_ + 1 < end ? p_ + 1 : end - 1, end, lane, nullptr); \
    o2 = idxwin_get<false>(win, a.col, nullptr, p_ + 2 < end ? p_ + 2 : end - 1, end, lane, nullptr); \
    o3 = idxwin_get<false>(win, a.col, nullptr, p_ + 3 < end ? p_ + 3 : end - 1, end, lane, nullptr); \
  } while (0)
  TailRegs<DH> ktn, vtn;
  constexpr int PF = DH == 32 ? AMPCONV_PF_FWD_T4 : 1;     // main tiles of the next PF edges in flight (ring slot = e % PF)
  MainRegs<DH> ring[PF];
  if (beg < end) {
    idxwin_load<false>(win, a.col, nullptr, beg, end, lane);
    AMPCONV_IDS(id0, id1, id2, id3, beg);
    tail_load<DH, FULL>(ktn, a.K, h, id0, id1, id2, id3, L, lane);
    main_load<DH, FULL>(ring[0], tile_ptr<const float>(a.K, id0, h), a.K.row_stride,
                        tile_ptr<const float>(a.V, id0, h), a.V.row_stride, L, lane);
    if (PF == 2 && beg + 1 < end)
      main_load<DH, FULL>(ring[PF - 1], tile_ptr<const float>(a.K, id1, h), a.K.row_stride,
                          tile_ptr<const float>(a.V, id1, h), a.V.row_stride, L, lane);
  }

  for (int p0 = beg; p0 < end; p0 += 4) {
    const bool more = p0 + 4 < end;
    if (more) AMPCONV_IDS(nid0, nid1, nid2, nid3, p0 + 4);
    const bool live = p0 + g < end;                    // lane group g <-> edge p0 + g of the tail tile

    // tail tile of the batch: S^T rows (edge g, token 16 + q), columns = destination tokens
    tail_to_lds<DH, FULL>(Ktail, ktn, L, lane);
    if (more) tail_load<DH, FULL>(ktn, a.K, h, nid0, nid1, nid2, nid3, L, lane);   // a whole batch ahead
    tail_load<DH, FULL>(vtn, a.V, h, id0, id1, id2, id3, L, lane);                  // this batch's closing operand
    __builtin_amdgcn_wave_barrier();
    f32x4 St[NTM];
    f32x4 St4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NTM; ++nt) St[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      float kt[C::KK];
      rowop_from_lds<DH>(kt, Ktail, 0, lane);
#pragma unroll
      for (int kk = 0; kk < C::KK; ++kk) {
#pragma unroll
        for (int nt = 0; nt < NTM; ++nt) St[nt] = MFMA16(kt[kk], qB[nt][kk], St[nt]);
        if constexpr (NT4) St4 = MFMA4(kt[kk], qT[kk], St4, 0);
      }
    }
    // NT4: lane (g, sg, jt) = tail score of (edge p0 + sg, source token 16 + g) against destination token 16 + jt
    float zt = 0.f, PtT = 0.f;
    if constexpr (NT4) {
      zt = reduce_transpose(St4);
      if (p0 + sg >= end || (!FULL && 16 + g >= L)) zt = kNegBig;
    }
    // per lane (edge g, destination column): tail max tm, u = exp2(S - tm) and their sum, once per
    // batch; an edge's main phase only rescales them by exp2(tm - m) / l.  (The destination pass keeps
    // the per-edge form: the four extra live registers spill inside its loop and the gain is lost.)
    float tm[NTM], tsum[NTM], csel[NTM];
#pragma unroll
    for (int nt = 0; nt < NTM; ++nt) {
      csel[nt] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (!live || (!FULL && 16 + q >= L)) St[nt][q] = kNegBig;
      tm[nt] = fmaxf(fmaxf(St[nt][0], St[nt][1]), fmaxf(St[nt][2], St[nt][3]));
#pragma unroll
      for (int q = 0; q < 4; ++q)
        St[nt][q] = (live && (FULL || 16 + q < L)) ? fast_exp2(St[nt][q] - tm[nt]) : 0.f;
      tsum[nt] = (St[nt][0] + St[nt][1]) + (St[nt][2] + St[nt][3]);
    }

#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (p0 + e >= end) break;
      main_to_lds<DH, FULL>(Kt, ring[e % PF], L, lane);
      {      // edge p0 + e + PF: ids of this batch, then of the next (clamped ids are never fetched: has_next)
        const bool has_next = p0 + e + PF < end;
        const int next = PF == 1 ? (e == 0 ? id1 : e == 1 ? id2 : e == 2 ? id3 : nid0)
                                 : (e == 0 ? id2 : e == 1 ? id3 : e == 2 ? nid0 : nid1);
        if (has_next)
          main_load<DH, FULL>(ring[e % PF], tile_ptr<const float>(a.K, next, h), a.K.row_stride,
                              tile_ptr<const float>(a.V, next, h), a.V.row_stride, L, lane);
      }
      __builtin_amdgcn_wave_barrier();
      f32x4 S[NTM];
      f32x4 S4 = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        float kA[C::KK];
        rowop_from_lds<DH>(kA, Kt, 0, lane);
#pragma unroll
        for (int nt = 0; nt < NTM; ++nt) S[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) {
#pragma unroll
          for (int nt = 0; nt < NTM; ++nt) S[nt] = MFMA16(kA[kk], qB[nt][kk], S[nt]);
          if constexpr (NT4) S4 = MFMA4(kA[kk], qT[kk], S4, 0);
        }
      }
      const bool mine = g == e;                          // this lane group holds edge e's tail rows
#pragma unroll
      for (int nt = 0; nt < NTM; ++nt) {
        f32x4 &s = S[nt];
        if (!FULL) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (4 * g + q >= L) s[q] = kNegBig;
        }
        float m = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
        m = groups_max(fmaxf(m, mine ? tm[nt] : kNegBig));
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = fast_exp2(s[q] - m);
        const float lm = (s[0] + s[1]) + (s[2] + s[3]);
        const float ct = fast_exp2(tm[nt] - m);            // meaningful in group e only
        const float inv = fast_rcp(groups_sum(lm + (mine ? ct * tsum[nt] : 0.f)));
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] *= inv;
        csel[nt] = mine ? ct * inv : csel[nt];             // edge e's tail rows: P^T = u * csel
      }
      float pz = 0.f;
      if constexpr (NT4) {        // columns 16..19: softmax over the 16 lanes that share jt (+ quad e of the tail tile)
        const bool mineT = sg == e;
        float z = reduce_transpose(S4);
        if (!FULL && 4 * sg + g >= L) z = kNegBig;
        const float m = groups_max(quads_max(fmaxf(z, mineT ? zt : kNegBig)));
        pz = fast_exp2(z - m);
        const float pt = mineT ? fast_exp2(zt - m) : 0.f;
        const float inv = fast_rcp(groups_sum(quads_sum(pz + pt)));
        pz *= inv;
        PtT = mineT ? pt * inv : PtT;
      }
#pragma unroll
      for (int mc = 0; mc < C::MC; ++mc) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float vA = Vt[lds_idx<DH>(4 * g + q, n + 16 * mc)];
#pragma unroll
          for (int nt = 0; nt < NTM; ++nt) OT[mc][nt] = MFMA16(vA, S[nt][q], OT[mc][nt]);
        }
      }
      if constexpr (NT4) nt_accumulate<DH, AMPCONV_NT_PHYS_FWD>(O4, Vt, nt_base, pz);
      __builtin_amdgcn_wave_barrier();
    }

    // closing product of the batch: contraction over (edge g, token 16 + q)
#pragma unroll
    for (int nt = 0; nt < NTM; ++nt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) St[nt][q] *= csel[nt];   // u = 0 for edges beyond the segment and tokens beyond L
    }
    tail_to_lds<DH, FULL>(Vtail, vtn, L, lane);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int mc = 0; mc < C::MC; ++mc) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float vA = Vtail[lds_idx<DH>(4 * g + q, n + 16 * mc)];
#pragma unroll
        for (int nt = 0; nt < NTM; ++nt) OT[mc][nt] = MFMA16(vA, St[nt][q], OT[mc][nt]);
      }
    }
    if constexpr (NT4) nt_accumulate<DH, AMPCONV_NT_PHYS_FWD>(O4, Vtail, nt_base, PtT);   // row 4 sg + x = (edge sg, token 16 + x)
    __builtin_amdgcn_wave_barrier();
    id0 = nid0; id1 = nid1; id2 = nid2; id3 = nid3;
  }
#undef AMPCONV_IDS

  const float inv = a.hub.mode == 2 ? 1.f : (deg > 0 ? 1.f / (float)deg : 0.f);
  float *ob = tile_ptr<float>(a.O, onode, h);
#pragma unroll
  for (int nt = 0; nt < NTM; ++nt) {
    const int i = n + 16 * nt;
    if (i < L) {
#pragma unroll
      for (int mc = 0; mc < C::MC; ++mc) {
        float4 o = make_float4(OT[mc][nt][0] * inv, OT[mc][nt][1] * inv, OT[mc][nt][2] * inv,
                               OT[mc][nt][3] * inv);
        *reinterpret_cast<float4 *>(ob + (int64_t)i * a.O.row_stride + 4 * g + 16 * mc) = o;
      }
    }
  }
  if constexpr (NT4) {          // the four sg partial sums of every block column, then one quad stores
    nt_fix_halves<DH, AMPCONV_NT_PHYS_FWD>(O4, lane);
#pragma unroll
    for (int hf = 0; hf < C::MC; ++hf) {
      float4 o;
      o.x = quads_sum(O4[hf][0]) * inv; o.y = quads_sum(O4[hf][1]) * inv;
      o.z = quads_sum(O4[hf][2]) * inv; o.w = quads_sum(O4[hf][3]) * inv;
      if (sg == 0 && 16 + jt < L)
        *reinterpret_cast<float4 *>(ob + (int64_t)(16 + jt) * a.O.row_stride + 16 * hf + 4 * g) = o;
    }
  }
}

// ---- backward, destination pass with the tail tokens of four consecutive edges in one MFMA tile
// (same batching as fwd_mfma_t4: the tail tiles S^T and dP^T of a batch come first, each edge's main
// phase folds its lane group of them into its column max / sum / delta, the batch closes with
// dQ^T += K_tail^T dS_tail^T over (edge, token) jointly).  84 -> 60 MFMAs per edge and head.
// The tail rows (K and V rows 16..19 of four edges) are needed from the first to the last phase of
// their batch, so the next batch's rows land in a second pair of LDS images, by LDS-DMA
// (global_load_lds_dwordx4: no staging registers; each lane fetches the 16-byte chunk that belongs
// at its linear position of the swizzled image).  They are issued right after the first main phase
// has consumed its staged registers and are complete by the time any younger ordinary load has
// been waited for, i.e. long before the next batch starts.
template <int DH, bool FULL>
__device__ __forceinline__ void tail_dma(float *img, const ampconv_view_t &view, int h, int id0, int id1, int id2,
                                         int id3, int L, int lane) {
  using C = TileCfg<DH>;
  constexpr int ROWS = 64 / C::CH;                 // image rows per wave-instruction (1 KiB)
#pragma unroll
  for (int i = 0; i < 16 / ROWS; ++i) {
    const int rho = ROWS * i + lane / C::CH, pos = lane % C::CH, tok = 16 + (rho & 3);
    const int lo = (rho & 4) ? id1 : id0, hi = (rho & 4) ? id3 : id2;
    const int node = (rho & 8) ? hi : lo;
    const float *src = tile_ptr<const float>(view, node, h) + (int64_t)tok * view.row_stride +
                       4 * (pos ^ swz<DH>(rho));
    // Issued as inline assembly on purpose: behind the builtin the compiler orders EVERY later LDS read
    // after the DMA (`s_waitcnt vmcnt(0)` right after the issue = the whole memory latency, once per
    // batch, also for reads of other LDS objects).  The caller waits for the images itself.
    const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)(img + ROWS * i * DH);
    unsigned saved_m0;
    if (FULL || tok < L)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(saved_m0)
                   : "v"(src), "s"(dst)
                   : "memory");
  }
}

// NT4: the destination's own tail tokens 16..19 (second column tile) on v_mfma_f32_4x4x1_16b_f32, as in
// fwd_mfma_t4: phase 1 leaves one register per product and 16-row tile (lane (g, sg, j) = source token
// 4 sg + g, destination token 16 + j); max, sum and delta of a column are all-reduces over the 16 lanes that
// share j, quad sg == e folding in the batch's tail tile.  Per edge and head 30 16x16x4 + 30 4x4x1 instead of 60.
template <int DH, bool FULL, bool STATS, bool NT4>
__global__ __launch_bounds__(64 * kWavesPerBlock, DH == 32 ? AMPCONV_DST_WAVES : AMPCONV_DST16_WAVES) void bwd_dst_mfma_t4(BwdArgs a) {
  using C = TileCfg<DH>;
  constexpr int kImg = 16 * DH;
  // two separate LDS objects: the compiler orders every LDS read behind an LDS-DMA it cannot prove
  // disjoint from it (a `vmcnt(0)` right after the issue, i.e. the whole memory latency once per
  // batch); reads of the main images provably never touch the tail images this way
  __shared__ __attribute__((aligned(16))) float lds_all[kWavesPerBlock][2 * kImg];
  __shared__ __attribute__((aligned(16))) float lds_tails[kWavesPerBlock][4 * kImg];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.n_units) return;
  int64_t r, onode;
  int h, beg, end, deg;
  if (!map_unit(a.hub, a.ptr, unit, a.n_units, a.H, r, onode, h, beg, end, deg)) return;
  constexpr int NTM = NT4 ? 1 : 2;           // 16-wide destination-token column tiles on the 16x16x4 path
  const int L = a.L, g = lane >> 4, n = lane & 15, sg = (lane >> 2) & 3, jt = lane & 3;
  float *Kt = lds_all[wave], *Vt = Kt + kImg, *tails = lds_tails[wave];
  const float inv = deg > 0 ? 1.f / (float)deg : 0.f;       // dO is the gradient of the MEAN
  const float oscale = a.hub.mode == 2 ? 1.f : a.oscale;

  float qB[NTM][C::KK], dOB[NTM][C::KK];
  float qT[NT4 ? C::KK : 1], gT[NT4 ? C::KK : 1];
  {
    const float *qb = tile_ptr<const float>(a.Q, r, h);
    const float *gb = tile_ptr<const float>(a.dO, r, h);
#pragma unroll
    for (int nt = 0; nt < NTM; ++nt) {
      rowop_from_global<DH>(qB[nt], qb, a.Q.row_stride, nt, true, a.qscale, L, lane);
      rowop_from_global<DH>(dOB[nt], gb, a.dO.row_stride, nt, true, inv, L, lane);
    }
    if constexpr (NT4) {
      tailop_from_global<DH>(qT, qb, a.Q.row_stride, a.qscale, L, lane);
      tailop_from_global<DH>(gT, gb, a.dO.row_stride, inv, L, lane);
    }
  }
  if (!FULL) {
    for (int i = lane; i < 2 * kImg; i += AMPCONV_WAVE) Kt[i] = 0.f;
    for (int i = lane; i < 4 * kImg; i += AMPCONV_WAVE) tails[i] = 0.f;
  }
  f32x4 dQT[C::MC][NTM];
  f32x4 dQ4[C::MC];                          // NT4: block (g, sg), reg rr, lane jt: channel 16 hf + 4 g + rr, token 16 + jt
#pragma unroll
  for (int mc = 0; mc < C::MC; ++mc) {
#pragma unroll
    for (int nt = 0; nt < NTM; ++nt) dQT[mc][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    dQ4[mc] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int nt_base = lds_idx<DH>(4 * sg, 4 * g + jt);        // NT4 phase 2: row 4 sg, channel 4 g + jt of a 16-row image

  IdxWindow win;
  const float *wts = reinterpret_cast<const float *>(a.spos);
  int id0 = 0, id1 = 0, id2 = 0, id3 = 0, nid0 = 0, nid1 = 0, nid2 = 0, nid3 = 0;
  float sp0 = 0.f, sp1 = 0.f, sp2 = 0.f, sp3 = 0.f, nsp0 = 0.f, nsp1 = 0.f, nsp2 = 0.f, nsp3 = 0.f;   // CSC positions (int bits)
#define AMPCONV_IDS(o0, o1, o2, o3, s0, s1, s2, s3, p)                                                 \
  do {                                                                                                 \
    const int p_ = (p);                                                                                \
    o0 = idxwin_get<STATS>(win, a.idx, wts, p_ < end ? p_ : end - 1, end, lane, &s0);                   \
    o1 = idxwin_get<STATS>(win, a.idx, wts, p_ + 1 < end ? p_ + 1 : end - 1, end, lane, &s1);           \
    o2 = idxwin_get<STATS>(win, a.idx, wts, p_ + 2 < end ? p_ + 2 : end - 1, end, lane, &s2);           \
    o3 = idxwin_get<STATS>(win, a.idx, wts, p_ + 3 < end ? p_ + 3 : end - 1, end, lane, &s3);           \
  } while (0)
  // main tiles of the next PF edges in flight (ring slot = e % PF).  The ordering argument for the LDS-DMA
  // of the tail images below holds for PF = 2 as well: the DMA of batch b + 1 goes out in phase 0 of the FULL
  // batch b ahead of that phase's staged loads, and phase 2 waits for exactly those loads.
  constexpr int PF = DH == 32 ? AMPCONV_PF_DST_T4 : AMPCONV_PF_DST16;
  MainRegs<DH> ring[PF];
  int cur = 0;
  if (beg < end) {
    idxwin_load<STATS>(win, a.idx, wts, beg, end, lane);
    AMPCONV_IDS(id0, id1, id2, id3, sp0, sp1, sp2, sp3, beg);
    __builtin_amdgcn_wave_barrier();
    tail_dma<DH, FULL>(tails, a.K, h, id0, id1, id2, id3, L, lane);
    tail_dma<DH, FULL>(tails + kImg, a.V, h, id0, id1, id2, id3, L, lane);
    main_load<DH, FULL>(ring[0], tile_ptr<const float>(a.K, id0, h), a.K.row_stride,
                        tile_ptr<const float>(a.V, id0, h), a.V.row_stride, L, lane);
    if (PF == 2 && beg + 1 < end)
      main_load<DH, FULL>(ring[PF - 1], tile_ptr<const float>(a.K, id1, h), a.K.row_stride,
                          tile_ptr<const float>(a.V, id1, h), a.V.row_stride, L, lane);
  }

  for (int p0 = beg; p0 < end; p0 += 4) {
    const bool more = p0 + 4 < end;
    if (more) AMPCONV_IDS(nid0, nid1, nid2, nid3, nsp0, nsp1, nsp2, nsp3, p0 + 4);
    const bool live = p0 + g < end;                    // lane group g <-> edge p0 + g of the tail tiles
    float *Ktail = tails + 2 * cur * kImg, *Vtail = Ktail + kImg;

    // this batch's tail images: the first batch's were issued in the prologue just BEFORE the staged
    // loads of the first main tile (everything but those youngest loads must have landed); a later
    // batch's were issued four main phases ago, and every main phase since has waited for loads
    // younger than them (vector-memory operations retire in order)
    if (p0 == beg) {
      if (!FULL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (MainRegs<DH>::NP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    __builtin_amdgcn_wave_barrier();
    f32x4 St[NTM], dPt[NTM];
    f32x4 St4 = f32x4{0.f, 0.f, 0.f, 0.f}, dPt4 = St4;
#pragma unroll
    for (int nt = 0; nt < NTM; ++nt) St[nt] = dPt[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      float kt[C::KK], vt[C::KK];
      rowop_from_lds<DH>(kt, Ktail, 0, lane);
      rowop_from_lds<DH>(vt, Vtail, 0, lane);
#pragma unroll
      for (int kk = 0; kk < C::KK; ++kk) {
#pragma unroll
        for (int nt = 0; nt < NTM; ++nt) {
          St[nt] = MFMA16(kt[kk], qB[nt][kk], St[nt]);
          dPt[nt] = MFMA16(vt[kk], dOB[nt][kk], dPt[nt]);
        }
        if constexpr (NT4) {
          St4 = MFMA4(kt[kk], qT[kk], St4, 0);
          dPt4 = MFMA4(vt[kk], gT[kk], dPt4, 0);
        }
      }
    }
    // NT4: lane (g, sg, jt) = (edge p0 + sg, source token 16 + g) against destination token 16 + jt
    float zt = 0.f, dzt = 0.f, DsT = 0.f;
    if constexpr (NT4) {
      zt = reduce_transpose(St4);
      dzt = reduce_transpose(dPt4);
      if (p0 + sg >= end || (!FULL && 16 + g >= L)) zt = kNegBig;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (!live || (!FULL && 16 + q >= L)) {
#pragma unroll
        for (int nt = 0; nt < NTM; ++nt) St[nt][q] = kNegBig;
      }
    float tmx[NTM];
#pragma unroll
    for (int nt = 0; nt < NTM; ++nt) tmx[nt] = fmaxf(fmaxf(St[nt][0], St[nt][1]), fmaxf(St[nt][2], St[nt][3]));

#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (p0 + e >= end) break;
      main_to_lds<DH, FULL>(Kt, ring[e % PF], L, lane);
      if (e == 0 && more) {          // next batch's tail rows -> the other pair of images; issued BEFORE
        float *Knext = tails + 2 * (cur ^ 1) * kImg;   // this phase's staged loads, so that the wait for
        tail_dma<DH, FULL>(Knext, a.K, h, nid0, nid1, nid2, nid3, L, lane);           // those (next phase)
        tail_dma<DH, FULL>(Knext + kImg, a.V, h, nid0, nid1, nid2, nid3, L, lane);    // does not stall on them
      }
      {
        const bool has_next = p0 + e + PF < end;
        const int next = PF == 1 ? (e == 0 ? id1 : e == 1 ? id2 : e == 2 ? id3 : nid0)
                                 : (e == 0 ? id2 : e == 1 ? id3 : e == 2 ? nid0 : nid1);
        if (has_next)
          main_load<DH, FULL>(ring[e % PF], tile_ptr<const float>(a.K, next, h), a.K.row_stride,
                              tile_ptr<const float>(a.V, next, h), a.V.row_stride, L, lane);
      }
      __builtin_amdgcn_wave_barrier();
      const bool mine = g == e;
      float *sb = nullptr;
      if (STATS) {
        const float pos = e == 0 ? sp0 : e == 1 ? sp1 : e == 2 ? sp2 : sp3;
        sb = a.stats + ((int64_t)__builtin_bit_cast(int, pos) * a.H + h) * kStatsPerUnit;
      }
      f32x4 S4 = f32x4{0.f, 0.f, 0.f, 0.f}, dP4 = S4;
#pragma unroll
      for (int nt = 0; nt < NTM; ++nt) {
        f32x4 S0 = f32x4{0.f, 0.f, 0.f, 0.f}, dP0 = S0;
        {
          float kA[C::KK], vA[C::KK];
          rowop_from_lds<DH>(kA, Kt, 0, lane);
          rowop_from_lds<DH>(vA, Vt, 0, lane);
#pragma unroll
          for (int kk = 0; kk < C::KK; ++kk) {
            S0 = MFMA16(kA[kk], qB[nt][kk], S0);
            dP0 = MFMA16(vA[kk], dOB[nt][kk], dP0);
            if constexpr (NT4) {
              S4 = MFMA4(kA[kk], qT[kk], S4, 0);
              dP4 = MFMA4(vA[kk], gT[kk], dP4, 0);
            }
          }
        }
        if (!FULL) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (4 * g + q >= L) S0[q] = kNegBig;
        }
        float m = fmaxf(fmaxf(S0[0], S0[1]), fmaxf(S0[2], S0[3]));
        m = groups_max(fmaxf(m, mine ? tmx[nt] : kNegBig));
        f32x4 pt;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          S0[q] = fast_exp2(S0[q] - m);
          pt[q] = fast_exp2(St[nt][q] - m);              // meaningful in group e only
        }
        const float lm = (S0[0] + S0[1]) + (S0[2] + S0[3]), lt = (pt[0] + pt[1]) + (pt[2] + pt[3]);
        const float l = groups_sum(lm + (mine ? lt : 0.f));
        const float rinv = fast_rcp(l);
        float part = 0.f, partt = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          S0[q] *= rinv;
          pt[q] *= rinv;
          part = fmaf(S0[q], dP0[q], part);
          partt = fmaf(pt[q], dPt[nt][q], partt);
        }
        const float delta = groups_sum(part + (mine ? partt : 0.f));
        if (STATS) {
          const int i = n + 16 * nt;
          if (g == 0 && i < kLmax) {
            sb[i] = m + __builtin_amdgcn_logf(l);
            sb[kLmax + i] = delta;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          S0[q] *= dP0[q] - delta;                                           // dS^T, main rows
          dPt[nt][q] = mine ? pt[q] * (dPt[nt][q] - delta) : dPt[nt][q];      // dS^T, edge e's tail rows
        }
#pragma unroll
        for (int mc = 0; mc < C::MC; ++mc) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float kC = Kt[lds_idx<DH>(4 * g + q, n + 16 * mc)];
            dQT[mc][nt] = MFMA16(kC, S0[q], dQT[mc][nt]);
          }
        }
      }
      if constexpr (NT4) {        // columns 16..19: all-reduces over the 16 lanes that share jt (+ quad e of the tail tile)
        const bool mineT = sg == e;
        float z = reduce_transpose(S4);
        const float dz = reduce_transpose(dP4);
        if (!FULL && 4 * sg + g >= L) z = kNegBig;
        const float m = groups_max(quads_max(fmaxf(z, mineT ? zt : kNegBig)));
        float pz = fast_exp2(z - m), pt = mineT ? fast_exp2(zt - m) : 0.f;
        const float l = groups_sum(quads_sum(pz + pt));
        const float rinv = fast_rcp(l);
        pz *= rinv;
        pt *= rinv;
        const float delta = groups_sum(quads_sum(fmaf(pz, dz, pt * dzt)));
        if (STATS) {
          if (lane < 4) {          // g == 0, sg == 0: destination token 16 + jt
            sb[16 + jt] = m + __builtin_amdgcn_logf(l);
            sb[kLmax + 16 + jt] = delta;
          }
        }
        DsT = mineT ? pt * (dzt - delta) : DsT;                 // dS^T of edge e's tail rows
        nt_accumulate<DH, AMPCONV_NT_PHYS_DST>(dQ4, Kt, nt_base, pz * (dz - delta));
      }
      __builtin_amdgcn_wave_barrier();
    }

    // closing product of the batch: contraction over (edge g, token 16 + q)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (!live) {
#pragma unroll
        for (int nt = 0; nt < NTM; ++nt) dPt[nt][q] = 0.f;
      }
#pragma unroll
    for (int mc = 0; mc < C::MC; ++mc) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float kC = Ktail[lds_idx<DH>(4 * g + q, n + 16 * mc)];
#pragma unroll
        for (int nt = 0; nt < NTM; ++nt) dQT[mc][nt] = MFMA16(kC, dPt[nt][q], dQT[mc][nt]);
      }
    }
    if constexpr (NT4) nt_accumulate<DH, AMPCONV_NT_PHYS_DST>(dQ4, Ktail, nt_base, DsT);   // row 4 sg + x = (edge sg, token 16 + x)
    __builtin_amdgcn_wave_barrier();
    id0 = nid0; id1 = nid1; id2 = nid2; id3 = nid3;
    sp0 = nsp0; sp1 = nsp1; sp2 = nsp2; sp3 = nsp3;
    cur ^= 1;
  }
#undef AMPCONV_IDS

  float *ob = tile_ptr<float>(a.dQ, onode, h);
  float wmax = 0.f;          // largest finite magnitude this lane stores (recorded at the end if the caller asked)
#pragma unroll
  for (int nt = 0; nt < NTM; ++nt) {
    const int i = n + 16 * nt;
    if (i < L) {
#pragma unroll
      for (int mc = 0; mc < C::MC; ++mc) {
        float4 o = make_float4(dQT[mc][nt][0] * oscale, dQT[mc][nt][1] * oscale,
                               dQT[mc][nt][2] * oscale, dQT[mc][nt][3] * oscale);
        *reinterpret_cast<float4 *>(ob + (int64_t)i * a.dQ.row_stride + 4 * g + 16 * mc) = o;
        if (kRecordAbsmax) wmax = finite_abs_max(wmax, o);
      }
    }
  }
  if constexpr (NT4) {          // the four sg partial sums of every block column, then one quad stores
    nt_fix_halves<DH, AMPCONV_NT_PHYS_DST>(dQ4, lane);
#pragma unroll
    for (int hf = 0; hf < C::MC; ++hf) {
      float4 o;
      o.x = quads_sum(dQ4[hf][0]) * oscale; o.y = quads_sum(dQ4[hf][1]) * oscale;
      o.z = quads_sum(dQ4[hf][2]) * oscale; o.w = quads_sum(dQ4[hf][3]) * oscale;
      if (sg == 0 && 16 + jt < L) {
        *reinterpret_cast<float4 *>(ob + (int64_t)(16 + jt) * a.dQ.row_stride + 16 * hf + 4 * g) = o;
        if (kRecordAbsmax) wmax = finite_abs_max(wmax, o);
      }
    }
  }
  if (kRecordAbsmax) {
    float *const amax_p = late_absmax_arg();
    if (amax_p) wave_record_absmax(amax_p, wmax);
  }
}

inline bool aligned16(const ampconv_view_t &v) {
  return ((uintptr_t)v.ptr % 16 == 0) && (v.node_stride % 4 == 0) && (v.row_stride % 4 == 0) &&
         (v.head_stride % 4 == 0);
}

}  // namespace

// This file is compiled into TWO objects (__graft_entry__.py): AMPCONV_PART=1 = the backward passes and the helpers
// under the default scheduler, AMPCONV_PART=2 = the forward pass under `-mllvm -amdgpu-sched-strategy=iterative-minreg`
// (the forward kernel sits exactly at the 128-register line of 4 waves per SIMD: the register-minimising scheduler
// needs no spill there and runs 4-5 % faster, 76-79 vs 80-82 ms at cfg4; the two backward kernels lose 2-3 % under it).
// AMPCONV_PART undefined: everything in one object.
#ifndef AMPCONV_PART
#define AMPCONV_PART 0
#endif

#if AMPCONV_PART != 2
bool ampconv_mfma_supported(int L, int D, int H) {
  const int dh = D / H;
  return L >= 1 && L <= kLmax && (dh == 16 || dh == 32);
}

bool ampconv_mfma_views_ok(const ampconv_view_t *views, int n) {
  for (int i = 0; i < n; ++i)
    if (!aligned16(views[i])) return false;
  return true;
}

#endif  // AMPCONV_PART != 2

#if AMPCONV_PART != 1
int ampconv_fwd_edge_mfma(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                          const int32_t *rowptr, const int32_t *col, const int32_t *qidx,
                          int64_t n_rows, int L, int D, int H, ampconv_view_t O, HubArgs hub,
                          hipStream_t stream) {
  const int dh = D / H;
  FwdArgs a{Q, K, V, O, rowptr, col, qidx, hub, n_rows * H, L, H, kLog2e / sqrtf((float)dh)};
  const int64_t blocks = (a.n_units + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > INT32_MAX) return AMPCONV_E_BADARG;
  const dim3 grid((unsigned)blocks), block(64 * kWavesPerBlock);
  static const bool t4 = !(std::getenv("AMPCONV_FWD_T4") && std::getenv("AMPCONV_FWD_T4")[0] == '0');
  static const bool nt4 = !(std::getenv("AMPCONV_FWD_NT4") && std::getenv("AMPCONV_FWD_NT4")[0] == '0');
  if (t4 && L > 16) {                    // batched tails pay only if there are tail tokens
    if (nt4) {                           // destination tail columns on 4x4x1
      if (dh == 32 && L == kLmax) fwd_mfma_t4<32, true, true><<<grid, block, 0, stream>>>(a);
      else if (dh == 32) fwd_mfma_t4<32, false, true><<<grid, block, 0, stream>>>(a);
      else if (L == kLmax) fwd_mfma_t4<16, true, true><<<grid, block, 0, stream>>>(a);
      else fwd_mfma_t4<16, false, true><<<grid, block, 0, stream>>>(a);
    } else {
      if (dh == 32 && L == kLmax) fwd_mfma_t4<32, true, false><<<grid, block, 0, stream>>>(a);
      else if (dh == 32) fwd_mfma_t4<32, false, false><<<grid, block, 0, stream>>>(a);
      else if (L == kLmax) fwd_mfma_t4<16, true, false><<<grid, block, 0, stream>>>(a);
      else fwd_mfma_t4<16, false, false><<<grid, block, 0, stream>>>(a);
    }
    return ampconv_launch_status();
  }
  if (dh == 32 && L == kLmax) fwd_mfma<32, true, AMPCONV_PF_FWD><<<grid, block, 0, stream>>>(a);
  else if (dh == 32) fwd_mfma<32, false, 1><<<grid, block, 0, stream>>>(a);
  else if (L == kLmax) fwd_mfma<16, true, AMPCONV_PF_FWD><<<grid, block, 0, stream>>>(a);
  else fwd_mfma<16, false, 1><<<grid, block, 0, stream>>>(a);
  return ampconv_launch_status();
}

#endif  // AMPCONV_PART != 1

#if AMPCONV_PART != 2
int ampconv_bwd_edge_dst_mfma(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                              ampconv_view_t dO, const int32_t *rowptr, const int32_t *col,
                              int64_t n_rows, int L, int D, int H, ampconv_view_t dQ, HubArgs hub,
                              StatsArgs st, hipStream_t stream) {
  const int dh = D / H;
  BwdArgs a{};
  a.hub = hub;
  a.spos = st.spos; a.stats = st.stats; a.absmax = st.absmax;
  a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.dQ = dQ;
  a.ptr = rowptr; a.idx = col; a.cinv = nullptr;
  a.n_units = n_rows * H; a.L = L; a.H = H;
  a.qscale = kLog2e / sqrtf((float)dh);
  a.oscale = 1.f / sqrtf((float)dh);
  const int64_t blocks = (a.n_units + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > INT32_MAX) return AMPCONV_E_BADARG;
  const dim3 grid((unsigned)blocks), block(64 * kWavesPerBlock);
  static const bool t4 = !(std::getenv("AMPCONV_DST_T4") && std::getenv("AMPCONV_DST_T4")[0] == '0');
  static const bool nt4 = !(std::getenv("AMPCONV_DST_NT4") && std::getenv("AMPCONV_DST_NT4")[0] == '0');
  if (t4 && L > 16) {                    // batched tails pay only if there are tail tokens
#define AMPCONV_DST_LAUNCH(DH_, FULL_)                                                                  \
  do {                                                                                                  \
    if (st.stats && nt4) bwd_dst_mfma_t4<DH_, FULL_, true, true><<<grid, block, 0, stream>>>(a);        \
    else if (st.stats) bwd_dst_mfma_t4<DH_, FULL_, true, false><<<grid, block, 0, stream>>>(a);         \
    else if (nt4) bwd_dst_mfma_t4<DH_, FULL_, false, true><<<grid, block, 0, stream>>>(a);              \
    else bwd_dst_mfma_t4<DH_, FULL_, false, false><<<grid, block, 0, stream>>>(a);                      \
  } while (0)
    if (dh == 32 && L == kLmax) AMPCONV_DST_LAUNCH(32, true);
    else if (dh == 32) AMPCONV_DST_LAUNCH(32, false);
    else if (L == kLmax) AMPCONV_DST_LAUNCH(16, true);
    else AMPCONV_DST_LAUNCH(16, false);
#undef AMPCONV_DST_LAUNCH
    return ampconv_launch_status();
  }
  if (st.stats) {
    if (dh == 32 && L == kLmax) bwd_dst_mfma<32, true, AMPCONV_PF_DST, true><<<grid, block, 0, stream>>>(a);
    else if (dh == 32) bwd_dst_mfma<32, false, 1, true><<<grid, block, 0, stream>>>(a);
    else if (L == kLmax) bwd_dst_mfma<16, true, AMPCONV_PF_DST, true><<<grid, block, 0, stream>>>(a);
    else bwd_dst_mfma<16, false, 1, true><<<grid, block, 0, stream>>>(a);
  } else {
    if (dh == 32 && L == kLmax) bwd_dst_mfma<32, true, AMPCONV_PF_DST, false><<<grid, block, 0, stream>>>(a);
    else if (dh == 32) bwd_dst_mfma<32, false, 1, false><<<grid, block, 0, stream>>>(a);
    else if (L == kLmax) bwd_dst_mfma<16, true, AMPCONV_PF_DST, false><<<grid, block, 0, stream>>>(a);
    else bwd_dst_mfma<16, false, 1, false><<<grid, block, 0, stream>>>(a);
  }
  return ampconv_launch_status();
}

int ampconv_bwd_edge_src_mfma(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                              ampconv_view_t dO, const int32_t *cscptr, const int32_t *crow,
                              const float *cinv, int64_t n_src, int L, int D, int H,
                              ampconv_view_t dK, ampconv_view_t dV, HubArgs hub, StatsArgs st,
                              hipStream_t stream) {
  const int dh = D / H;
  BwdArgs a{};
  a.hub = hub;
  a.stats = st.stats; a.absmax = st.absmax;
  a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.dK = dK; a.dV = dV;
  a.ptr = cscptr; a.idx = crow; a.cinv = cinv;
  a.n_units = n_src * H; a.L = L; a.H = H;
  a.qscale = kLog2e / sqrtf((float)dh);
  a.oscale = 0.6931471805599453f;     // dK = ln2 * sum dS^T (Q * log2e / sqrt(dh))
  const int64_t blocks = (a.n_units + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > INT32_MAX) return AMPCONV_E_BADARG;
  const dim3 grid((unsigned)blocks), block(64 * kWavesPerBlock);
  static const bool t4 = !(std::getenv("AMPCONV_SRC_T4") && std::getenv("AMPCONV_SRC_T4")[0] == '0');
  static const bool nt4 = !(std::getenv("AMPCONV_SRC_NT4") && std::getenv("AMPCONV_SRC_NT4")[0] == '0');
  // developer knob: unused dynamic LDS per block, to lower the occupancy in experiments
  static const unsigned dyn = std::getenv("AMPCONV_SRC_DYNLDS") ? (unsigned)atoi(std::getenv("AMPCONV_SRC_DYNLDS")) : 0u;
  if (st.stats && t4 && nt4 && L > 16) {       // tail columns on 4x4x1 (only if there are tail tokens)
    if (dh == 32 && L == kLmax) bwd_src_mfma_t4<32, true, true><<<grid, block, dyn, stream>>>(a);
    else if (dh == 32) bwd_src_mfma_t4<32, false, true><<<grid, block, dyn, stream>>>(a);
    else if (L == kLmax) bwd_src_mfma_t4<16, true, true><<<grid, block, dyn, stream>>>(a);
    else bwd_src_mfma_t4<16, false, true><<<grid, block, dyn, stream>>>(a);
  } else if (st.stats && t4) {
    if (dh == 32 && L == kLmax) bwd_src_mfma_t4<32, true, false><<<grid, block, dyn, stream>>>(a);
    else if (dh == 32) bwd_src_mfma_t4<32, false, false><<<grid, block, dyn, stream>>>(a);
    else if (L == kLmax) bwd_src_mfma_t4<16, true, false><<<grid, block, dyn, stream>>>(a);
    else bwd_src_mfma_t4<16, false, false><<<grid, block, dyn, stream>>>(a);
  } else if (st.stats) {
    if (dh == 32 && L == kLmax) bwd_src_mfma<32, true, AMPCONV_PF_SRC, true><<<grid, block, dyn, stream>>>(a);
    else if (dh == 32) bwd_src_mfma<32, false, 1, true><<<grid, block, dyn, stream>>>(a);
    else if (L == kLmax) bwd_src_mfma<16, true, AMPCONV_PF_SRC, true><<<grid, block, dyn, stream>>>(a);
    else bwd_src_mfma<16, false, 1, true><<<grid, block, dyn, stream>>>(a);
  } else {
    if (dh == 32 && L == kLmax) bwd_src_mfma<32, true, AMPCONV_PF_SRC, false><<<grid, block, dyn, stream>>>(a);
    else if (dh == 32) bwd_src_mfma<32, false, 1, false><<<grid, block, dyn, stream>>>(a);
    else if (L == kLmax) bwd_src_mfma<16, true, AMPCONV_PF_SRC, false><<<grid, block, dyn, stream>>>(a);
    else bwd_src_mfma<16, false, 1, false><<<grid, block, dyn, stream>>>(a);
  }
  return ampconv_launch_status();
}

#endif  // AMPCONV_PART != 2

#if defined(AMPCONV_STAMPS) && AMPCONV_PART != 2
extern "C" int ampconv_debug_read_stamps(unsigned long long *host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamp_sums), sizeof(unsigned long long) * n);
}
#endif
