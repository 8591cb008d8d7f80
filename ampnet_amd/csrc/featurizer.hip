// AMPGCN featuriser on the device ("next" row 2 of SURVEY.md section 8f): the step right before
// the first AMPConv layer.  Reference: src/ampnet/module/amp_gcn.py:120-183
//   :122-125  x_ = StandardScaler().fit_transform(x)            (per-feature z-score over nodes)
//   :132-135  per node: sample L of its PRESENT (non-zero) features, with replacement
//   :146-147  token = cat(feature_embedding_table.weight[f], x_[node, f])   -> [L, De + 1]
//   :152-153  flatten to [N, L * D]
// The reference does this in a per-node Python loop with np.random.choice; its random stream
// cannot be matched, so sampling uses this library's counter-based generator (uniform over the
// present features, with replacement) and everything that is a function of the sampled indices is
// checked exactly against a numpy/sklearn restatement (oracle/featurizer_numpy.py).
#include "common.h"

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// one thread per feature column: mean and 1/std (population variance, constant columns -> scale 1,
// sklearn StandardScaler semantics), double accumulation
__global__ void zscore_stats_kernel(const float *__restrict__ x, int64_t N, int64_t F,
                                    float *__restrict__ mean, float *__restrict__ inv_std) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  double s = 0.0;
  for (int64_t n = 0; n < N; ++n) s += (double)x[n * F + f];
  const double m = s / (double)N;
  double v = 0.0;
  for (int64_t n = 0; n < N; ++n) {
    const double d = (double)x[n * F + f] - m;
    v += d * d;
  }
  v /= (double)N;
  mean[f] = (float)m;
  // sklearn's _is_constant_feature: a variance at rounding-noise level means a constant feature
  const double eps = 2.220446049250313e-16, nm = (double)N * m * eps;
  const bool constant = v <= (double)N * eps * v + nm * nm;
  inv_std[f] = constant ? 1.f : (float)(1.0 / sqrt(v));
}

// one wavefront per node: idx[n, l] = a uniformly random present feature (with replacement);
// nodes with no present feature get -1 and raise the `empty` flag (np.random.choice raises there)
__global__ __launch_bounds__(64) void sample_present_kernel(const float *__restrict__ x, int64_t N, int F,
                                                            int L, uint64_t seed, int32_t *__restrict__ idx,
                                                            int32_t *__restrict__ empty) {
  extern __shared__ unsigned long long masks[];      // ceil(F/64) ballot masks, then prefix counts
  const int64_t n = blockIdx.x;
  const int lane = threadIdx.x;
  const int nchunk = (F + 63) / 64;
  int *prefix = reinterpret_cast<int *>(masks + nchunk);
  const float *row = x + n * (int64_t)F;
  int total = 0;
  for (int c = 0; c < nchunk; ++c) {
    const int f = c * 64 + lane;
    const unsigned long long m = __ballot(f < F && row[f] != 0.f);
    if (lane == 0) {
      masks[c] = m;
      prefix[c] = total;
    }
    total += __popcll(m);
  }
  __syncthreads();
  if (total == 0) {
    if (lane == 0) atomicOr(empty, 1);
    for (int l = lane; l < L; l += 64) idx[n * L + l] = -1;
    return;
  }
  for (int l = lane; l < L; l += 64) {
    const uint64_t r = splitmix64(seed ^ splitmix64((uint64_t)n * 0x100000001B3ull + (uint64_t)l));
    int k = (int)(r % (uint64_t)total);               // rank among the present features
    int c = 0;
    while (c + 1 < nchunk && prefix[c + 1] <= k) ++c;
    k -= prefix[c];
    unsigned long long m = masks[c];
    for (int t = 0; t < k; ++t) m &= m - 1;            // drop the k lowest set bits
    idx[n * L + l] = c * 64 + __ffsll((long long)m) - 1;
  }
}

// out[n, l, :De] = table[idx[n,l], :], out[n, l, De] = (x[n, idx] - mean) * inv_std
__global__ void build_tokens_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                    const float *__restrict__ inv_std, const int32_t *__restrict__ idx,
                                    const float *__restrict__ table, int64_t NL, int F, int L, int De,
                                    float *__restrict__ out) {
  const int64_t t = blockIdx.x;                        // token (n, l)
  const int f = idx[t];
  const int64_t n = t / L;
  float *o = out + t * (int64_t)(De + 1);
  if (f < 0) {
    for (int c = threadIdx.x; c <= De; c += blockDim.x) o[c] = 0.f;
    return;
  }
  for (int c = threadIdx.x; c < De; c += blockDim.x) o[c] = table[(int64_t)f * De + c];
  if (threadIdx.x == 0) o[De] = (x[n * (int64_t)F + f] - mean[f]) * inv_std[f];
}

// dtable[f, :] += dout[n, l, :De] for every token with idx == f (float atomics: order of the adds,
// hence the last bits, may differ run to run -- as with torch's own embedding backward)
__global__ void table_grad_kernel(const float *__restrict__ dout, const int32_t *__restrict__ idx, int De,
                                  float *__restrict__ dtable) {
  const int64_t t = blockIdx.x;
  const int f = idx[t];
  if (f < 0) return;
  const float *g = dout + t * (int64_t)(De + 1);
  for (int c = threadIdx.x; c < De; c += blockDim.x) atomicAdd(dtable + (int64_t)f * De + c, g[c]);
}

}  // namespace

extern "C" int ampconv_feat_zscore_stats(const float *x, int64_t N, int64_t F, float *mean, float *inv_std,
                                         void *stream) {
  if (!x || !mean || !inv_std || N <= 0 || F <= 0) return AMPCONV_E_BADARG;
  zscore_stats_kernel<<<(unsigned)((F + 63) / 64), 64, 0, (hipStream_t)stream>>>(x, N, F, mean, inv_std);
  return ampconv_launch_status();
}

extern "C" int ampconv_feat_sample_present(const float *x, int64_t N, int F, int L, uint64_t seed,
                                           int32_t *idx, int32_t *empty_flag, void *stream) {
  if (!x || !idx || !empty_flag || N <= 0 || F <= 0 || L <= 0 || N > INT32_MAX) return AMPCONV_E_BADARG;
  hipError_t e = hipMemsetAsync(empty_flag, 0, sizeof(int32_t), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  const int nchunk = (F + 63) / 64;
  const size_t lds = (size_t)nchunk * (sizeof(unsigned long long) + sizeof(int));
  sample_present_kernel<<<(unsigned)N, 64, lds, (hipStream_t)stream>>>(x, N, F, L, seed, idx, empty_flag);
  return ampconv_launch_status();
}

extern "C" int ampconv_feat_build(const float *x, const float *mean, const float *inv_std, const int32_t *idx,
                                  const float *table, int64_t N, int F, int L, int De, float *out,
                                  void *stream) {
  if (!x || !mean || !inv_std || !idx || !table || !out || N <= 0 || F <= 0 || L <= 0 || De < 0)
    return AMPCONV_E_BADARG;
  if (N * L > INT32_MAX) return AMPCONV_E_BADARG;
  build_tokens_kernel<<<(unsigned)(N * L), 64, 0, (hipStream_t)stream>>>(x, mean, inv_std, idx, table, N * L, F,
                                                                        L, De, out);
  return ampconv_launch_status();
}

extern "C" int ampconv_feat_table_grad(const float *dout, const int32_t *idx, int64_t N, int L, int De, int F,
                                       float *dtable, void *stream) {
  if (!dout || !idx || !dtable || N <= 0 || L <= 0 || De <= 0 || F <= 0 || N * L > INT32_MAX)
    return AMPCONV_E_BADARG;
  hipError_t e = hipMemsetAsync(dtable, 0, sizeof(float) * (size_t)F * De, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  table_grad_kernel<<<(unsigned)(N * L), 64, 0, (hipStream_t)stream>>>(dout, idx, De, dtable);
  return ampconv_launch_status();
}
