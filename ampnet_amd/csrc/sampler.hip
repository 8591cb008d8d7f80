// GraphSAINT random-walk sampler on the GPU ("next" row 1 of SURVEY.md section 8f): the step that
// feeds the hot path in the reference's training harness
// (experiments/cora_benchmark_graphsaint.py:80-82,96-106).  In-tree spec = the reference's
// vendored copy of PyG's sampler, visualization/visualize_graphsaint_subgraphs.py:
//   :195-199  __sample_nodes__: start = randint(N, batch_size); random_walk(start, walk_length)
//   :107-110  __getitem__: unique (sorted) walked nodes + saint_subgraph (induced sub-graph)
//   :112-135  __collate__: relabelled edge_index, node/edge attribute subsets, norms
//   :137-173  __compute_norm__: node/edge occurrence counts -> node_norm, edge_norm
// The random stream itself cannot be matched (torch_sparse's random_walk is not in the image):
// walks use a counter-based generator (splitmix64 of (seed, walk, step)), so a (seed, graph)
// pair always yields the same batch; everything after the walk is deterministic and is checked
// against a numpy restatement (oracle/graphsaint_numpy.py).
#include <hipcub/hipcub.hpp>
#include "common.h"

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// one thread per walk: walks[b, 0] = start[b]; step t picks a uniform out-neighbour of the current
// node (CSC by source: cscptr/crow), or stays if it has none
__global__ void random_walk_kernel(const int32_t *__restrict__ cscptr, const int32_t *__restrict__ crow,
                                   const int64_t *__restrict__ start, int64_t B, int walk_length,
                                   uint64_t seed, int64_t *__restrict__ walks) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int64_t cur = start[b];
  int64_t *w = walks + b * (walk_length + 1);
  w[0] = cur;
  for (int t = 0; t < walk_length; ++t) {
    const int beg = cscptr[cur], deg = cscptr[cur + 1] - beg;
    if (deg > 0) {
      const uint64_t r = splitmix64(seed ^ splitmix64((uint64_t)b * 0x100000001B3ull + (uint64_t)t));
      cur = crow[beg + (int)(r % (uint64_t)deg)];
    }
    w[t + 1] = cur;
  }
}

__global__ void mark_kernel(const int64_t *__restrict__ nodes, int64_t n, int32_t *__restrict__ mark) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) mark[nodes[i]] = 1;
}

// relabel = exclusive scan of mark; node_idx[relabel[v]] = v for marked v (ascending = sorted unique)
__global__ void compact_nodes_kernel(const int32_t *__restrict__ mark, const int32_t *__restrict__ relabel,
                                     int64_t N, int64_t *__restrict__ node_idx, int32_t *__restrict__ n_sub) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= N) return;
  if (mark[v]) node_idx[relabel[v]] = v;
  if (v == N - 1) *n_sub = relabel[v] + mark[v];
}

// kept out-edges of the k-th sampled node (both endpoints sampled)
__global__ void count_edges_kernel(const int64_t *__restrict__ node_idx, int64_t n_sub,
                                   const int32_t *__restrict__ cscptr, const int32_t *__restrict__ crow,
                                   const int32_t *__restrict__ mark, int32_t *__restrict__ cnt) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_sub) return;
  const int64_t u = node_idx[k];
  int c = 0;
  for (int p = cscptr[u]; p < cscptr[u + 1]; ++p) c += mark[crow[p]];
  cnt[k] = c;
}

// the same with the number of sampled nodes still on the device: k runs over an upper bound, entries past it count 0
__global__ void count_edges_bounded_kernel(const int64_t *__restrict__ node_idx, int64_t n_bound,
                                           const int32_t *__restrict__ n_sub_dev, const int32_t *__restrict__ cscptr,
                                           const int32_t *__restrict__ crow, const int32_t *__restrict__ mark,
                                           int32_t *__restrict__ cnt) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k > n_bound) return;
  int c = 0;
  if (k < *n_sub_dev) {
    const int64_t u = node_idx[k];
    for (int p = cscptr[u]; p < cscptr[u + 1]; ++p) c += mark[crow[p]];
  }
  cnt[k] = c;
}

__global__ void fill_edges_kernel(const int64_t *__restrict__ node_idx, int64_t n_sub,
                                  const int32_t *__restrict__ cscptr, const int32_t *__restrict__ crow,
                                  const int32_t *__restrict__ cperm, const int32_t *__restrict__ mark,
                                  const int32_t *__restrict__ relabel, const int32_t *__restrict__ off,
                                  int64_t E_sub, int64_t *__restrict__ edge_index,
                                  int64_t *__restrict__ edge_id) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_sub) return;
  const int64_t u = node_idx[k];
  int64_t o = off[k];
  for (int p = cscptr[u]; p < cscptr[u + 1]; ++p) {
    const int v = crow[p];
    if (mark[v]) {
      edge_index[o] = k;                     // source, relabelled
      edge_index[E_sub + o] = relabel[v];    // destination, relabelled
      edge_id[o] = cperm[p];                 // original edge id
      ++o;
    }
  }
}

__global__ void add_counts_kernel(const int64_t *__restrict__ idx, int64_t n, float *__restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(count + idx[i], 1.0f);      // integer-valued, order independent
}

// visualize_graphsaint_subgraphs.py:165-171
__global__ void norms_kernel(const float *__restrict__ node_count, const float *__restrict__ edge_count,
                             const int64_t *__restrict__ edge_src, int64_t N, int64_t E, float num_samples,
                             float *__restrict__ node_norm, float *__restrict__ edge_norm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < E) {
    const float t = node_count[edge_src[i]], c = edge_count[i];
    // the reference divides first: 0/0 -> NaN -> 0.1, x/0 -> inf -> clamp 1e4 (counts are >= 0)
    edge_norm[i] = c == 0.f ? (t == 0.f ? 0.1f : 1e4f) : fminf(t / c, 1e4f);
  }
  if (i < N) {
    const float c = node_count[i] == 0.f ? 0.1f : node_count[i];
    node_norm[i] = num_samples / c / (float)N;
  }
}

size_t scan_bytes(int64_t n) {
  size_t t = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t, (const int32_t *)nullptr, (int32_t *)nullptr, (int)n,
                                         (hipStream_t)0);
  return t;
}

}  // namespace

extern "C" int ampconv_saint_random_walk(const int32_t *cscptr, const int32_t *crow, const int64_t *start,
                                         int64_t B, int walk_length, uint64_t seed, int64_t *walks,
                                         void *stream) {
  if (!cscptr || !start || !walks || B < 0 || walk_length < 0) return AMPCONV_E_BADARG;
  if (B == 0) return AMPCONV_OK;
  random_walk_kernel<<<(unsigned)((B + 127) / 128), 128, 0, (hipStream_t)stream>>>(cscptr, crow, start, B,
                                                                                  walk_length, seed, walks);
  return ampconv_launch_status();
}

extern "C" size_t ampconv_saint_workspace_bytes(int64_t N) {
  return N > 0 ? scan_bytes(N + 1) + 256 : 0;
}

// nodes[n] (with repeats) -> mark[N] (0/1), relabel[N] (exclusive scan), node_idx (sorted unique,
// capacity N), n_sub (device int32)
extern "C" int ampconv_saint_nodes(const int64_t *nodes, int64_t n, int64_t N, int32_t *mark,
                                   int32_t *relabel, int64_t *node_idx, int32_t *n_sub, void *workspace,
                                   size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!nodes || !mark || !relabel || !node_idx || !n_sub || !workspace || N <= 0 || N > INT32_MAX || n < 0)
    return AMPCONV_E_BADARG;
  if (workspace_bytes < ampconv_saint_workspace_bytes(N)) return AMPCONV_E_WORKSPACE;
  hipError_t e = hipMemsetAsync(mark, 0, sizeof(int32_t) * N, stream);
  if (e != hipSuccess) return (int)e;
  if (n > 0) mark_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(nodes, n, mark);
  size_t tb = scan_bytes(N);
  e = hipcub::DeviceScan::ExclusiveSum(workspace, tb, mark, relabel, (int)N, stream);
  if (e != hipSuccess) return (int)e;
  compact_nodes_kernel<<<(unsigned)((N + 255) / 256), 256, 0, stream>>>(mark, relabel, N, node_idx, n_sub);
  return ampconv_launch_status();
}

// per sampled node: number of kept out-edges -> cnt[n_sub], off[n_sub] (exclusive scan),
// e_sub (device int32) = total
extern "C" int ampconv_saint_count_edges(const int64_t *node_idx, int64_t n_sub, const int32_t *cscptr,
                                         const int32_t *crow, const int32_t *mark, int32_t *cnt,
                                         int32_t *off, int32_t *e_sub, void *workspace,
                                         size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_sub < 0 || !e_sub) return AMPCONV_E_BADARG;
  if (n_sub == 0) {
    hipError_t e = hipMemsetAsync(e_sub, 0, sizeof(int32_t), stream);
    return e == hipSuccess ? AMPCONV_OK : (int)e;
  }
  if (!node_idx || !cscptr || !mark || !cnt || !off || !workspace) return AMPCONV_E_BADARG;
  if (workspace_bytes < scan_bytes(n_sub + 1)) return AMPCONV_E_WORKSPACE;
  // cnt has n_sub + 1 entries, the last one 0, so that off[n_sub] = total
  hipError_t e = hipMemsetAsync(cnt + n_sub, 0, sizeof(int32_t), stream);
  if (e != hipSuccess) return (int)e;
  count_edges_kernel<<<(unsigned)((n_sub + 127) / 128), 128, 0, stream>>>(node_idx, n_sub, cscptr, crow, mark,
                                                                          cnt);
  size_t tb = scan_bytes(n_sub + 1);
  e = hipcub::DeviceScan::ExclusiveSum(workspace, tb, cnt, off, (int)(n_sub + 1), stream);
  if (e != hipSuccess) return (int)e;
  e = hipMemcpyAsync(e_sub, off + n_sub, sizeof(int32_t), hipMemcpyDeviceToDevice, stream);
  return e == hipSuccess ? AMPCONV_OK : (int)e;
}

// as count_edges, before the host knows n_sub: `n_bound` >= n_sub (e.g. min(walked nodes, N)), n_sub read from the
// device; cnt / off have n_bound + 1 entries, e_sub = off[n_bound].  Lets ONE read-back return both sizes.
extern "C" int ampconv_saint_count_edges_bounded(const int64_t *node_idx, int64_t n_bound, const int32_t *n_sub_dev,
                                                 const int32_t *cscptr, const int32_t *crow, const int32_t *mark,
                                                 int32_t *cnt, int32_t *off, int32_t *e_sub, void *workspace,
                                                 size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_bound < 0 || !e_sub || !n_sub_dev || !node_idx || !cscptr || !mark || !cnt || !off || !workspace)
    return AMPCONV_E_BADARG;
  if (workspace_bytes < scan_bytes(n_bound + 1)) return AMPCONV_E_WORKSPACE;
  count_edges_bounded_kernel<<<(unsigned)((n_bound + 1 + 127) / 128), 128, 0, stream>>>(node_idx, n_bound, n_sub_dev,
                                                                                      cscptr, crow, mark, cnt);
  size_t tb = scan_bytes(n_bound + 1);
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(workspace, tb, cnt, off, (int)(n_bound + 1), stream);
  if (e != hipSuccess) return (int)e;
  e = hipMemcpyAsync(e_sub, off + n_bound, sizeof(int32_t), hipMemcpyDeviceToDevice, stream);
  return e == hipSuccess ? AMPCONV_OK : (int)e;
}

extern "C" int ampconv_saint_fill_edges(const int64_t *node_idx, int64_t n_sub, const int32_t *cscptr,
                                        const int32_t *crow, const int32_t *cperm, const int32_t *mark,
                                        const int32_t *relabel, const int32_t *off, int64_t E_sub,
                                        int64_t *edge_index, int64_t *edge_id, void *stream) {
  if (n_sub < 0 || E_sub < 0) return AMPCONV_E_BADARG;
  if (n_sub == 0 || E_sub == 0) return AMPCONV_OK;
  if (!node_idx || !cscptr || !crow || !cperm || !mark || !relabel || !off || !edge_index || !edge_id)
    return AMPCONV_E_BADARG;
  fill_edges_kernel<<<(unsigned)((n_sub + 127) / 128), 128, 0, (hipStream_t)stream>>>(
      node_idx, n_sub, cscptr, crow, cperm, mark, relabel, off, E_sub, edge_index, edge_id);
  return ampconv_launch_status();
}

extern "C" int ampconv_saint_add_counts(const int64_t *idx, int64_t n, float *count, void *stream) {
  if (n < 0) return AMPCONV_E_BADARG;
  if (n == 0) return AMPCONV_OK;
  if (!idx || !count) return AMPCONV_E_BADARG;
  add_counts_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(idx, n, count);
  return ampconv_launch_status();
}

extern "C" int ampconv_saint_norms(const float *node_count, const float *edge_count, const int64_t *edge_src,
                                   int64_t N, int64_t E, float num_samples, float *node_norm,
                                   float *edge_norm, void *stream) {
  if (N <= 0 || E < 0 || !node_count || !node_norm) return AMPCONV_E_BADARG;
  if (E > 0 && (!edge_count || !edge_src || !edge_norm)) return AMPCONV_E_BADARG;
  const int64_t n = N > E ? N : E;
  norms_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(node_count, edge_count, edge_src,
                                                                            N, E, num_samples, node_norm,
                                                                            edge_norm);
  return ampconv_launch_status();
}

// ---- the batch's feature rows: dst[i, :] = src[idx[i], :], 16 bytes per lane, one workgroup per 4 KiB piece of a row
namespace {
__global__ __launch_bounds__(256) void gather_rows_kernel(const char *__restrict__ src, int64_t stride, int64_t row_bytes,
                                                          const int64_t *__restrict__ idx, int64_t n, char *__restrict__ dst,
                                                          int pieces) {
  const int64_t u = blockIdx.x;
  const int64_t i = u / pieces;
  const int64_t off = (u - i * pieces) * 4096 + (int64_t)threadIdx.x * 16;
  if (i >= n || off >= row_bytes) return;
  const int64_t r = idx[i];
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  const i32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const i32x4_t *>(src + r * stride + off));
  __builtin_nontemporal_store(v, reinterpret_cast<i32x4_t *>(dst + i * row_bytes + off));
}
}  // namespace

extern "C" int ampconv_saint_gather_rows(const void *src, int64_t src_stride_bytes, int64_t row_bytes, const int64_t *idx,
                                         int64_t n, void *dst, void *stream) {
  if (n < 0 || row_bytes < 0) return AMPCONV_E_BADARG;
  if (n == 0 || row_bytes == 0) return AMPCONV_OK;
  if (!src || !idx || !dst || row_bytes % 16 || src_stride_bytes % 16 || (uintptr_t)src % 16 || (uintptr_t)dst % 16)
    return AMPCONV_E_BADARG;
  const int pieces = (int)((row_bytes + 4095) / 4096);
  const int64_t nb = n * pieces;
  if (nb > INT32_MAX) return AMPCONV_E_BADARG;
  gather_rows_kernel<<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>((const char *)src, src_stride_bytes, row_bytes, idx, n,
                                                                    (char *)dst, pieces);
  return ampconv_launch_status();
}
