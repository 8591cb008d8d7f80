// Graph preparation: edge_index [2, E] int64 -> dst-sorted CSR + src-sorted CSC.
//
// Replaces the implicit index_select / scatter bookkeeping of PyG's
// MessagePassing.propagate (reference src/ampnet/conv/amp_conv.py:25).  Both
// sorts are STABLE radix sorts (hipcub), so the order in which the edge kernels
// add the contributions of one destination / one source is fixed by the input
// and results are bitwise reproducible run to run.
#include <hipcub/hipcub.hpp>
#include "common.h"

namespace {

constexpr size_t kAlign = 256;
inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

__global__ void extract_keys(const int64_t *__restrict__ idx, int64_t E, int64_t N,
                             int32_t *__restrict__ keys, int32_t *__restrict__ vals,
                             int32_t *__restrict__ oob) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= E) return;
  int64_t v = idx[p];
  if (v < 0 || v >= N) {
    atomicOr(oob, 1);
    v = v < 0 ? 0 : N - 1;
  }
  keys[p] = (int32_t)v;
  vals[p] = (int32_t)p;
}

// other[p] = clamp(idx_other[perm[p]])
__global__ void gather_other(const int32_t *__restrict__ perm, const int64_t *__restrict__ idx_other,
                             int64_t E, int64_t N, int32_t *__restrict__ other) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= E) return;
  int64_t v = idx_other[perm[p]];
  v = v < 0 ? 0 : (v >= N ? N - 1 : v);
  other[p] = (int32_t)v;
}

// cinv[p] = 1 / in-degree of the destination of CSC edge p
__global__ void edge_inv_degree(const int32_t *__restrict__ crow, const int32_t *__restrict__ rowptr,
                                int64_t E, float *__restrict__ cinv) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= E) return;
  const int d = crow[p];
  cinv[p] = 1.f / (float)(rowptr[d + 1] - rowptr[d]);
}

// ptr[n] = first sorted position whose key is >= n (lower bound); ptr[N] = E
__global__ void fill_ptr(const int32_t *__restrict__ sorted_keys, int64_t E, int64_t N,
                         int32_t *__restrict__ ptr) {
  int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n > N) return;
  int64_t lo = 0, hi = E;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (sorted_keys[mid] < n) lo = mid + 1; else hi = mid;
  }
  ptr[n] = (int32_t)lo;
}

int bits_for(int64_t N) {
  int b = 1;
  while (b < 31 && ((int64_t)1 << b) < N) ++b;
  return b;
}

size_t cub_temp_bytes(int64_t E, int64_t N) {
  size_t t = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t, (const int32_t *)nullptr, (int32_t *)nullptr,
                                     (const int32_t *)nullptr, (int32_t *)nullptr, (int)E, 0,
                                     bits_for(N), (hipStream_t)0);
  return t;
}

// CSR position -> CSC position of the same edge (two passes through the original edge ids)
__global__ void scatter_positions(const int32_t *__restrict__ cperm, int64_t E, int32_t *__restrict__ by_edge) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < E) by_edge[cperm[q]] = (int32_t)q;
}
__global__ void gather_positions(const int32_t *__restrict__ eperm, const int32_t *__restrict__ by_edge,
                                 int64_t E, int32_t *__restrict__ spos) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < E) spos[p] = by_edge[eperm[p]];
}

}  // namespace

extern "C" size_t ampconv_csr_workspace_bytes(int64_t N, int64_t E) {
  if (N <= 0 || E < 0) return 0;
  size_t e = align_up((size_t)(E > 0 ? E : 1) * sizeof(int32_t));
  return 3 * e + align_up(cub_temp_bytes(E > 0 ? E : 1, N)) + kAlign;
}

extern "C" int ampconv_csr_build(const int64_t *edge_index, int64_t E, int64_t N, int32_t *rowptr,
                                 int32_t *col, int32_t *eperm, int32_t *cscptr, int32_t *crow,
                                 int32_t *cperm, float *cinv, int32_t *oob, void *workspace,
                                 size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N <= 0 || E < 0 || N > INT32_MAX || E >= INT32_MAX) return AMPCONV_E_BADARG;
  if (!rowptr || !cscptr || !oob) return AMPCONV_E_BADARG;
  if (E > 0 && (!edge_index || !col || !eperm || !crow || !cperm || !workspace))
    return AMPCONV_E_BADARG;
  if (E > 0 && workspace_bytes < ampconv_csr_workspace_bytes(N, E)) return AMPCONV_E_WORKSPACE;

  hipError_t err = hipMemsetAsync(oob, 0, sizeof(int32_t), stream);
  if (err != hipSuccess) return (int)err;

  const int T = 256;
  const int gridn = (int)((N + 1 + T - 1) / T);
  if (E == 0) {
    fill_ptr<<<gridn, T, 0, stream>>>(nullptr, 0, N, rowptr);
    fill_ptr<<<gridn, T, 0, stream>>>(nullptr, 0, N, cscptr);
    return ampconv_launch_status();
  }

  char *ws = (char *)workspace;
  ws = (char *)align_up((size_t)ws);
  size_t e = align_up((size_t)E * sizeof(int32_t));
  int32_t *keys_in = (int32_t *)ws;
  int32_t *keys_out = (int32_t *)(ws + e);
  int32_t *vals_in = (int32_t *)(ws + 2 * e);
  void *cub_tmp = ws + 3 * e;
  size_t cub_bytes = cub_temp_bytes(E, N);
  const int grid = (int)((E + T - 1) / T);
  const int nbits = bits_for(N);

  const int64_t *src = edge_index;
  const int64_t *dst = edge_index + E;

  // dst-sorted CSR
  extract_keys<<<grid, T, 0, stream>>>(dst, E, N, keys_in, vals_in, oob);
  err = hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_bytes, keys_in, keys_out, vals_in, eperm,
                                           (int)E, 0, nbits, stream);
  if (err != hipSuccess) return (int)err;
  gather_other<<<grid, T, 0, stream>>>(eperm, src, E, N, col);
  fill_ptr<<<gridn, T, 0, stream>>>(keys_out, E, N, rowptr);

  // src-sorted CSC
  extract_keys<<<grid, T, 0, stream>>>(src, E, N, keys_in, vals_in, oob);
  err = hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_bytes, keys_in, keys_out, vals_in, cperm,
                                           (int)E, 0, nbits, stream);
  if (err != hipSuccess) return (int)err;
  gather_other<<<grid, T, 0, stream>>>(cperm, dst, E, N, crow);
  fill_ptr<<<gridn, T, 0, stream>>>(keys_out, E, N, cscptr);
  if (cinv) edge_inv_degree<<<grid, T, 0, stream>>>(crow, rowptr, E, cinv);
  return ampconv_launch_status();
}

extern "C" int ampconv_csc_positions(const int32_t *eperm, const int32_t *cperm, int64_t E, int32_t *scratch,
                                     int32_t *spos, void *stream) {
  if (E < 0 || E > INT32_MAX) return AMPCONV_E_BADARG;
  if (E == 0) return AMPCONV_OK;
  if (!eperm || !cperm || !scratch || !spos) return AMPCONV_E_BADARG;
  const int T = 256;
  const int grid = (int)((E + T - 1) / T);
  scatter_positions<<<grid, T, 0, (hipStream_t)stream>>>(cperm, E, scratch);
  gather_positions<<<grid, T, 0, (hipStream_t)stream>>>(eperm, scratch, E, spos);
  return ampconv_launch_status();
}
