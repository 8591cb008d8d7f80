// Shared helpers for the libampconv.so HIP sources (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ampconv.h"

#define AMPCONV_WAVE 64

static inline int ampconv_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? AMPCONV_OK : (int)e;
}

static inline bool view_ok(const ampconv_view_t &v) { return v.ptr != nullptr; }

// element address of (node n, token l, head h) channel 0
template <typename T>
__device__ __forceinline__ T *tile_ptr(const ampconv_view_t &v, int64_t n, int h) {
  return reinterpret_cast<T *>(v.ptr) + n * v.node_stride + (int64_t)h * v.head_stride;
}

__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}
// |x|, or 0 for NaN / infinity (operand maxima skip non-finite values: include/ampconv.h "SCALED MODE")
__device__ __forceinline__ float finite_abs(float x) {
  const float a = __builtin_fabsf(x);
  return a < __builtin_inff() ? a : 0.f;
}
__device__ __forceinline__ float finite_abs_max(float m, const float4 &v) {
  return fmaxf(fmaxf(m, fmaxf(finite_abs(v.x), finite_abs(v.y))), fmaxf(finite_abs(v.z), finite_abs(v.w)));
}
// one wave's share of a running maximum kept in *p (non-negative floats order as their bits): a plain read first -- a
// stale value only costs an atomic that changes nothing -- so that almost every wave of a launch leaves it at the read
__device__ __forceinline__ void wave_record_absmax(float *p, float m) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > __builtin_nontemporal_load(p))
    atomicMax(reinterpret_cast<unsigned *>(p), __builtin_bit_cast(unsigned, m));
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

// ---- long segments ("hubs", BASELINE config 5): a CSR/CSC segment longer than `chunk` edges is
// cut into chunks that separate waves reduce into partial tiles; an ordered combine pass adds
// the partial tiles of one row (deterministic).  Descriptors are built by ampconv_hub_plan.
struct HubDesc {
  int32_t row, beg, end;   // row of the segment, CSR/CSC positions [beg, end) of this chunk
  int32_t nfirst;          // number of chunks of the row if this is its first chunk, else 0
};
// softmax statistics handed from the destination pass to the source pass: per (CSC position, head)
// 20 log2-sum-exp values then 20 delta values (one per destination token; include/ampconv.h)
constexpr int kStatsPerUnit = 40;
struct StatsArgs {
  const int32_t *spos;     // CSR position -> CSC position (dst pass only)
  float *stats;            // nullptr: the source pass reduces its own softmax
  float *absmax;           // nullptr, or: atomic max of the finite magnitudes the pass writes (include/ampconv.h;
                           // recorded by edge_mfma's destination pass and the combine pass, the C entry covers the rest)
};

struct HubArgs {
  const int32_t *header;   // plan header {n_chunks, chunk, 0, 0}; descriptors follow at header + 4
  int mode;                // 0: no plan, 1: main pass (rows longer than header[1] are skipped),
                           // 2: hub pass (one unit per (chunk, head), writes partial tiles)
};

// Workgroup-per-unit kernels: blockIdx -> unit such that the H heads of one row run on ONE XCD, back to back.
// Workgroups go to the 8 XCDs round-robin, so with units in (row, head) order the heads of a row land on H different
// XCDs: where a head's slice of a token row is shorter than a 128-byte line (dh = 50 fp32: 200 bytes at 8-byte alignment
// = 2.6 lines) every L2 fetches the neighbours' bytes again.  Block b = 8 s + x (x = XCD, s = its slot
// there) takes row 8 (s / H) + x, head s % H.  Launch xcd_grid() workgroups; returns -1 for the padding.
__device__ __forceinline__ int64_t xcd_unit(int64_t b, int64_t n_units, int H) {
  const int x = (int)(b & 7);
  const int64_t s = b >> 3, rs = s / H;
  const int h = (int)(s - rs * H);
  const int64_t u = rs * 8 + x;
  return u * H + h < n_units ? u * H + h : -1;
}
static inline int64_t xcd_grid(int64_t n_units, int H) { return ((n_units / H + 7) / 8) * 8 * H; }

// unit -> (row, head, edge range); returns false if this wave has nothing to do.
// Main pass of a graph WITH a long-segment plan (mode 1, i.e. a skewed degree distribution): the row order is a
// pseudo-random permutation of the rows (scramble_row).  Workgroups go to the 8 XCDs round-robin, so with rows in
// id order XCD x only ever sees node ids with (2 * id) mod 8 in {x, x - 1}; on an R-MAT graph the degree follows the
// id's bits (every 0 bit multiplies the expected degree by ~3: ids ending in 00 have ~10 x the edges of ids ending
// in 11), one XCD gets most of the work and the launch lasts as long as that XCD.  Any assignment that looks at a
// few id bits only has the same problem; the permutation mixes all of them.
// Bijection on [0, n): two xorshift-multiply rounds on k = bit_length(n - 1) bits (each step is invertible on k-bit
// integers), cycle-walked back into [0, n) (fewer than two rounds on average since 2^k < 2 n).
__device__ __forceinline__ int64_t scramble_row(int64_t u, int64_t n) {
  if (n < 2) return u;
  const int k = 64 - __builtin_clzll((unsigned long long)(n - 1));
  const unsigned mask = k >= 32 ? 0xFFFFFFFFu : ((1u << k) - 1u);
  const int hs = (k + 1) >> 1;
  unsigned x = (unsigned)u;
  do {
    x ^= x >> hs;
    x = (x * 0x9E3779B1u) & mask;
    x ^= x >> hs;
    x = (x * 0x85EBCA6Bu) & mask;
    x ^= x >> hs;
  } while ((int64_t)x >= n);
  return (int64_t)x;
}
__device__ __forceinline__ bool map_unit(const HubArgs &hub, const int32_t *ptr, int64_t unit, int64_t n_units, int H,
                                         int64_t &row, int64_t &out_node, int &h, int &beg, int &end,
                                         int &deg) {
  int64_t u = unit / H;
  h = (int)(unit - u * H);
  if (hub.mode == 2) {
    const HubDesc d = reinterpret_cast<const HubDesc *>(hub.header + 4)[u];
    row = d.row;
    out_node = u;
    beg = d.beg;
    end = d.end;
    deg = ptr[row + 1] - ptr[row];
    return true;
  }
  if (hub.mode == 1) u = scramble_row(u, n_units / H);
  row = out_node = u;
  beg = ptr[u];
  end = ptr[u + 1];
  deg = end - beg;
  return !(hub.mode == 1 && deg > hub.header[1]);
}

// ---- internal entry points (edge_generic.hip / edge_mfma.hip), dispatched by edge_api.hip
int ampconv_fwd_edge_generic(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                             const int32_t *rowptr, const int32_t *col, const int32_t *qidx,
                             int64_t n_rows, int L, int D, int H, ampconv_view_t O,
                             hipStream_t stream);
int ampconv_bwd_edge_dst_generic(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                                 ampconv_view_t dO, const int32_t *rowptr, const int32_t *col,
                                 int64_t n_rows, int L, int D, int H, ampconv_view_t dQ,
                                 hipStream_t stream);
int ampconv_bwd_edge_src_generic(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                                 ampconv_view_t dO, const int32_t *cscptr, const int32_t *crow,
                                 const float *cinv, int64_t n_src, int L, int D, int H,
                                 ampconv_view_t dK, ampconv_view_t dV, hipStream_t stream);

// ---- MFMA fast path (edge_mfma.hip): L <= 20, dh in {16, 32}, 16-byte aligned views
bool ampconv_mfma_supported(int L, int D, int H);
bool ampconv_mfma_views_ok(const ampconv_view_t *views, int n);
int ampconv_fwd_edge_mfma(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                          const int32_t *rowptr, const int32_t *col, const int32_t *qidx,
                          int64_t n_rows, int L, int D, int H, ampconv_view_t O, HubArgs hub,
                          hipStream_t stream);
int ampconv_bwd_edge_dst_mfma(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                              ampconv_view_t dO, const int32_t *rowptr, const int32_t *col,
                              int64_t n_rows, int L, int D, int H, ampconv_view_t dQ, HubArgs hub,
                              StatsArgs st, hipStream_t stream);
int ampconv_bwd_edge_src_mfma(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                              ampconv_view_t dO, const int32_t *cscptr, const int32_t *crow,
                              const float *cinv, int64_t n_src, int L, int D, int H,
                              ampconv_view_t dK, ampconv_view_t dV, HubArgs hub, StatsArgs st,
                              hipStream_t stream);

// ---- workgroup-per-unit MFMA path (edge_block.hip): L <= 64, even dh <= 64, views aligned to two elements; fp32 or
// bf16 storage (`bf16`), fp32 arithmetic.  The source pass of this path exists only with the statistics of the
// destination pass.
bool ampconv_block_supported(int L, int D, int H, const ampconv_view_t *views, int n, bool bf16);
int ampconv_block_stats_floats(int L);      // floats per (edge, head): 2 * 16 * ceil(L / 16)
int ampconv_fwd_edge_block(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, const int32_t *rowptr,
                           const int32_t *col, const int32_t *qidx, int64_t n_rows, int L, int D, int H,
                           ampconv_view_t O, HubArgs hub, bool bf16, hipStream_t stream);
int ampconv_bwd_edge_dst_block(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dO,
                               const int32_t *rowptr, const int32_t *col, int64_t n_rows, int L, int D, int H,
                               ampconv_view_t dQ, HubArgs hub, StatsArgs sa, bool bf16, hipStream_t stream);
int ampconv_bwd_edge_src_block(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dO,
                               const int32_t *cscptr, const int32_t *crow, const float *cinv, int64_t n_src,
                               int L, int D, int H, ampconv_view_t dK, ampconv_view_t dV, HubArgs hub,
                               const float *stats, bool bf16, hipStream_t stream);

// ---- the same path on the 16-bit matrix pipe (edge_block_x3.hip: fp32 rows as three bf16 planes per value, six partial
// products; bf16 rows as they lie, one product); `vec` = vec_of() of the views (4 or 2 elements)
bool ampconv_block_x3_supported(int L, int D, int H, bool bf16);
int ampconv_fwd_edge_block_x3(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, const int32_t *rowptr,
                              const int32_t *col, const int32_t *qidx, int64_t n_rows, int L, int D, int H,
                              ampconv_view_t O, HubArgs hub, int vec, bool bf16, hipStream_t stream);
int ampconv_bwd_edge_dst_block_x3(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dO,
                                  const int32_t *rowptr, const int32_t *col, int64_t n_rows, int L, int D, int H,
                                  ampconv_view_t dQ, HubArgs hub, StatsArgs sa, int vec, bool bf16, hipStream_t stream);
int ampconv_bwd_edge_src_block_x3(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dO,
                                  const int32_t *cscptr, const int32_t *crow, const float *cinv, int64_t n_src, int L,
                                  int D, int H, ampconv_view_t dK, ampconv_view_t dV, HubArgs hub, const float *stats,
                                  int vec, bool bf16, hipStream_t stream);

// ---- short token sequences (edge_small.hip): L <= 4, one wave per row, VALU only, fp32; views aligned to the lane's
// vector width.  No softmax statistics.
bool ampconv_small_supported(int L, int D, int H, const ampconv_view_t *views, int n);
int ampconv_fwd_edge_small(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, const int32_t *rowptr,
                           const int32_t *col, const int32_t *qidx, int64_t n_rows, int L, int D, int H,
                           ampconv_view_t O, HubArgs hub, hipStream_t stream);
int ampconv_bwd_edge_dst_small(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dO,
                               const int32_t *rowptr, const int32_t *col, int64_t n_rows, int L, int D, int H,
                               ampconv_view_t dQ, HubArgs hub, hipStream_t stream);
int ampconv_bwd_edge_src_small(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dO,
                               const int32_t *cscptr, const int32_t *crow, const float *cinv, int64_t n_src,
                               int L, int D, int H, ampconv_view_t dK, ampconv_view_t dV, HubArgs hub,
                               hipStream_t stream);

// hub.hip
int ampconv_hub_combine(const void *plan, int64_t n_chunks, const float *P, ampconv_view_t out,
                        const int32_t *ptr_for_mean, int L, int D, int H, float scale, int out_bf16,
                        hipStream_t stream, float *absmax = nullptr);

// ---- bf16-storage path (edge_mfma_bf16.hip): L <= 20, dh == 32, 16-byte aligned bf16 views
bool ampconv_bf16_supported(int L, int D, int H, const ampconv_view_t *views, int n);
int ampconv_fwd_edge_bf16(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, const int32_t *rowptr,
                          const int32_t *col, const int32_t *qidx, int64_t n_rows, int L, int D, int H,
                          ampconv_view_t O, HubArgs hub, hipStream_t stream);
int ampconv_bwd_edge_dst_bf16(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dO,
                              const int32_t *rowptr, const int32_t *col, int64_t n_rows, int L, int D,
                              int H, ampconv_view_t dQ, HubArgs hub, hipStream_t stream);
int ampconv_bwd_edge_src_bf16(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dO,
                              const int32_t *cscptr, const int32_t *crow, const float *cinv,
                              int64_t n_src, int L, int D, int H, ampconv_view_t dK, ampconv_view_t dV,
                              HubArgs hub, hipStream_t stream);
