// Graph preparation: edge_index [2, E] int64 -> dst-sorted CSR + src-sorted CSC.
//
// Replaces the implicit index_select / scatter bookkeeping of PyG's
// MessagePassing.propagate (reference src/ampnet/conv/amp_conv.py:25).  Both
// sorts are STABLE radix sorts (hipcub), so the order in which the edge kernels
// add the contributions of one destination / one source is fixed by the input
// and results are bitwise reproducible run to run.
#include <stdlib.h>
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


// ---- small graphs (GraphSAINT batches, Cora: the reference's actual regime): the whole preparation -- both stable
// sorts, both pointer arrays, 1 / in-degree per CSC edge, the bounds flag and both long-segment plans -- in ONE launch
// of two workgroups (0: destination-sorted CSR, 1: source-sorted CSC) instead of ~26 (the device-wide radix sort
// alone is 5 launches per direction at this size, each a few microseconds of work).  Stable by construction: the
// sort key is (node id << 14 | edge id), sorted by a workgroup-wide radix sort in LDS.
constexpr int kSmallT = 1024, kSmallItems = 12, kSmallEbits = 14;
constexpr int kSmallMaxE = kSmallT * kSmallItems;       // 12 288 edges (Cora: 10 556)
constexpr int kSmallMaxN = 1 << 14;

struct SmallArgs {
  const int64_t *ei;
  int E, N, nbits, chunk, max_chunks;
  int32_t *ptr[2], *oth[2], *perm[2];
  float *cinv;
  int32_t *oob;
  int32_t *plan[2];
  int32_t *by_edge;           // CSC position of every original edge id (workgroup 1), or null
};

__global__ __launch_bounds__(kSmallT) void csr_small_kernel(SmallArgs a) {
  using Sort = hipcub::BlockRadixSort<uint32_t, kSmallT, kSmallItems>;
  __shared__ union {
    typename Sort::TempStorage sort;
    uint32_t sorted[kSmallMaxE];
  } u;
  __shared__ int deg[kSmallMaxN];          // workgroup 1: in-degree of every node
  const int b = blockIdx.x, t = threadIdx.x;
  const int64_t *key_arr = a.ei + (b == 0 ? a.E : 0);       // destinations for the CSR, sources for the CSC
  const int64_t *oth_arr = a.ei + (b == 0 ? 0 : a.E);
  const int64_t *dst_arr = a.ei + a.E;
  uint32_t ck[kSmallItems];
  bool bad = false;
#pragma unroll
  for (int i = 0; i < kSmallItems; ++i) {
    const int e = t * kSmallItems + i;
    ck[i] = 0xFFFFFFFFu;
    if (e < a.E) {
      int64_t v = key_arr[e];
      if (v < 0 || v >= a.N) {
        bad = true;
        v = v < 0 ? 0 : a.N - 1;
      }
      ck[i] = ((uint32_t)v << kSmallEbits) | (uint32_t)e;
    }
  }
  if (b == 1) {
    for (int n = t; n < a.N; n += kSmallT) deg[n] = 0;
    __syncthreads();
    for (int e = t; e < a.E; e += kSmallT) {
      int64_t d = dst_arr[e];
      d = d < 0 ? 0 : (d >= a.N ? a.N - 1 : d);
      atomicAdd(&deg[d], 1);
    }
  }
  if (bad) atomicOr(a.oob, 1);
  __syncthreads();
  // LSD radix passes are stable and the edge ids (low bits) ascend in the blocked input: sorting the node-id bits
  // alone leaves every segment in original edge order
  Sort(u.sort).Sort(ck, kSmallEbits, kSmallEbits + a.nbits);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kSmallItems; ++i) u.sorted[t * kSmallItems + i] = ck[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kSmallItems; ++i) {
    const int p = t * kSmallItems + i;
    if (p < a.E) {
      const int eid = (int)(ck[i] & ((1u << kSmallEbits) - 1));
      int64_t o = oth_arr[eid];
      o = o < 0 ? 0 : (o >= a.N ? a.N - 1 : o);
      a.perm[b][p] = eid;
      a.oth[b][p] = (int32_t)o;
      if (b == 1 && a.cinv) a.cinv[p] = 1.f / (float)deg[o];
      if (b == 1 && a.by_edge) a.by_edge[eid] = p;
    }
  }
  int32_t *ptr = a.ptr[b];
  for (int n = t; n <= a.N; n += kSmallT) {        // ptr[n] = first sorted position whose node id is >= n
    const uint32_t want = (uint32_t)n << kSmallEbits;
    int lo = 0, hi = a.E;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (u.sorted[mid] < want) lo = mid + 1; else hi = mid;
    }
    ptr[n] = lo;
  }
  if (!a.plan[b]) {
    if (t == 0) a.oob[1 + b] = 0;
    return;
  }
  // long-segment plan of this direction (same layout as hub_plan_kernel, csrc/hub.hip)
  int32_t *header = a.plan[b];
  HubDesc *descs = reinterpret_cast<HubDesc *>(header + 4);
  int32_t *firsts = reinterpret_cast<int32_t *>(descs + a.max_chunks);
  if (t == 0) {
    header[0] = 0;
    header[1] = a.chunk;
    header[2] = 0;
    header[3] = (int32_t)(firsts - header);
  }
  __syncthreads();                                   // ptr[] and the header, written above, are visible to the workgroup
  for (int r = t; r < a.N; r += kSmallT) {
    const int beg = ptr[r], end = ptr[r + 1];
    const int dg = end - beg;
    if (dg <= a.chunk) continue;
    const int nc = (dg + a.chunk - 1) / a.chunk;
    const int base = atomicAdd(header, nc);          // slot order is arbitrary, results do not depend on it
    firsts[atomicAdd(header + 2, 1)] = base;
    for (int k = 0; k < nc; ++k) {
      const int bb = beg + k * a.chunk;
      descs[base + k] = HubDesc{(int32_t)r, bb, bb + a.chunk < end ? bb + a.chunk : end, k == 0 ? nc : 0};
    }
  }
  __syncthreads();
  if (t == 0) a.oob[1 + b] = header[0];              // chunk count beside the bounds flag: one read-back for the host
}

// node n is listed iff it has an in-edge (which & 1) or an out-edge (which & 2)
__global__ void active_flags_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ cscptr, int64_t N,
                                    int which, int32_t *__restrict__ flag) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n > N) return;
  bool on = false;
  if (n < N) {
    if (which & 1) on = rowptr[n + 1] != rowptr[n];
    if (which & 2) on = on || cscptr[n + 1] != cscptr[n];
  }
  flag[n] = on ? 1 : 0;
}

__global__ void active_fill_kernel(const int32_t *__restrict__ ptr, int64_t N, int32_t *__restrict__ list,
                                   int32_t *__restrict__ count) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N && ptr[n + 1] != ptr[n]) list[ptr[n]] = (int32_t)n;
  if (n < 8) list[ptr[N] + n] = 0;                 // the padding the projection kernels read past the end (not used)
  if (n == 0) *count = ptr[N];
}

__global__ void plan_counts_kernel(const int32_t *plan_dst, const int32_t *plan_src, int32_t *oob) {
  oob[1] = plan_dst ? plan_dst[0] : 0;
  oob[2] = plan_src ? plan_src[0] : 0;
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

// hub.hip
int64_t ampconv_hub_max_chunks(int64_t E, int chunk);

extern "C" int ampconv_graph_build(const int64_t *edge_index, int64_t E, int64_t N, int32_t *rowptr, int32_t *col,
                                   int32_t *eperm, int32_t *cscptr, int32_t *crow, int32_t *cperm, float *cinv,
                                   int32_t *status, int chunk, void *plan_dst, void *plan_src, int32_t *by_edge,
                                   void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const bool plans = chunk > 0 && plan_dst && plan_src;
  static const bool no_small = [] {
    const char *e = getenv("AMPCONV_CSR_SMALL");
    return e && e[0] == '0';
  }();
  if (!status) return AMPCONV_E_BADARG;
  hipError_t err = hipMemsetAsync(status, 0, 4 * sizeof(int32_t), stream);
  if (err != hipSuccess) return (int)err;
  if (!no_small && E > 0 && E <= kSmallMaxE && N > 0 && N <= kSmallMaxN) {
    if (!edge_index || !rowptr || !col || !eperm || !cscptr || !crow || !cperm) return AMPCONV_E_BADARG;
    SmallArgs a{edge_index, (int)E, (int)N, bits_for(N), chunk, plans ? (int)ampconv_hub_max_chunks(E, chunk) : 0,
                {rowptr, cscptr}, {col, crow}, {eperm, cperm}, cinv, status,
                {plans ? (int32_t *)plan_dst : nullptr, plans ? (int32_t *)plan_src : nullptr}, by_edge};
    csr_small_kernel<<<2, kSmallT, 0, stream>>>(a);
    return ampconv_launch_status();
  }
  if (int rc = ampconv_csr_build(edge_index, E, N, rowptr, col, eperm, cscptr, crow, cperm, cinv, status, workspace,
                                 workspace_bytes, stream_))
    return rc;
  if (E > 0 && by_edge) scatter_positions<<<(unsigned)((E + 255) / 256), 256, 0, stream>>>(cperm, E, by_edge);
  if (plans && E > 0) {
    if (int rc = ampconv_hub_plan(rowptr, N, E, chunk, plan_dst, stream_)) return rc;
    if (int rc = ampconv_hub_plan(cscptr, N, E, chunk, plan_src, stream_)) return rc;
    plan_counts_kernel<<<1, 1, 0, stream>>>((const int32_t *)plan_dst, (const int32_t *)plan_src, status);
  }
  return ampconv_launch_status();
}

extern "C" int ampconv_csc_positions_from(const int32_t *eperm, const int32_t *by_edge, int64_t E, int32_t *spos,
                                          void *stream) {
  if (E < 0 || E > INT32_MAX) return AMPCONV_E_BADARG;
  if (E == 0) return AMPCONV_OK;
  if (!eperm || !by_edge || !spos) return AMPCONV_E_BADARG;
  gather_positions<<<(unsigned)((E + 255) / 256), 256, 0, (hipStream_t)stream>>>(eperm, by_edge, E, spos);
  return ampconv_launch_status();
}

extern "C" size_t ampconv_active_nodes_workspace_bytes(int64_t N) {
  if (N <= 0 || N >= INT32_MAX) return 0;
  size_t tmp = 0;
  hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, (int32_t *)nullptr, (int32_t *)nullptr, (int)(N + 1));
  return align_up((size_t)(N + 1) * sizeof(int32_t)) + align_up(tmp) + kAlign;
}

extern "C" int ampconv_active_nodes(const int32_t *rowptr, const int32_t *cscptr, int64_t N, int which, int32_t *list,
                                    int32_t *ptr, int32_t *count, void *workspace, size_t workspace_bytes,
                                    void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N <= 0 || N >= INT32_MAX || which < 1 || which > 3) return AMPCONV_E_BADARG;
  if (((which & 1) && !rowptr) || ((which & 2) && !cscptr) || !list || !ptr || !count || !workspace)
    return AMPCONV_E_BADARG;
  if (workspace_bytes < ampconv_active_nodes_workspace_bytes(N)) return AMPCONV_E_WORKSPACE;
  char *ws = (char *)align_up((size_t)workspace);
  int32_t *flag = (int32_t *)ws;
  void *tmp = ws + align_up((size_t)(N + 1) * sizeof(int32_t));
  size_t tmp_bytes = 0;
  hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, flag, ptr, (int)(N + 1));
  const unsigned grid = (unsigned)((N + 1 + 255) / 256);
  active_flags_kernel<<<grid, 256, 0, stream>>>(rowptr, cscptr, N, which, flag);
  hipError_t err = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, flag, ptr, (int)(N + 1), stream);
  if (err != hipSuccess) return (int)err;
  active_fill_kernel<<<grid, 256, 0, stream>>>(ptr, N, list, count);
  return ampconv_launch_status();
}
