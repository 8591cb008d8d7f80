// Long-segment ("hub") plan and the ordered combine of partial tiles (include/ampconv.h,
// "long segments").  Replaces nothing in the reference: PyG's scatter handles any degree; here
// it keeps one-wave-per-(row, head) kernels from running as long as their longest segment
// (BASELINE config 5: RMAT, in-degrees up to ~1e5).
#include "common.h"

namespace {

// Plan layout: header {n_chunks, chunk, n_rows, offset of the list in int32 units}, then max_chunks descriptors, then the list of the first-chunk
// indices of the long rows (n_rows entries; what the combine pass launches over: one block column per ROW instead of
// one per chunk with all but the first exiting -- 8 M blocks per call on RMAT scale 21).
__device__ __host__ inline int64_t hub_max_chunks(int64_t E, int chunk) { return 2 * (E / chunk) + 2; }

__global__ void hub_plan_kernel(const int32_t *__restrict__ ptr, int64_t N, int chunk,
                                int32_t *__restrict__ header, HubDesc *__restrict__ descs,
                                int32_t *__restrict__ firsts) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r == 0) {
    header[1] = chunk;
    header[3] = (int32_t)(firsts - header);
  }
  if (r >= N) return;
  const int beg = ptr[r], end = ptr[r + 1];
  const int deg = end - beg;
  if (deg <= chunk) return;
  const int nc = (deg + chunk - 1) / chunk;
  const int base = atomicAdd(header, nc);      // slot order is arbitrary, results do not depend on it
  firsts[atomicAdd(header + 2, 1)] = base;
  for (int k = 0; k < nc; ++k) {
    const int b = beg + k * chunk;
    descs[base + k] = HubDesc{(int32_t)r, b, b + chunk < end ? b + chunk : end, k == 0 ? nc : 0};
  }
}

// out[row] = scale * sum_{k < n} P[c + k] for the row whose first chunk is c.
// Grid (chunk, slab): the block of a row's FIRST chunk and slab s owns elements [VEC * 64 * s, VEC * 64 * (s + 1))
// of the L x D tile; its 256 threads are 64 element lanes x 4 chunk phases: phase q adds the chunks
// k = q, q + 4, ... on two alternating accumulators (8 independent loads in flight per lane), the four
// phase sums meet in LDS and are added in the fixed order ((p0 + p1) + (p2 + p3)).  A fixed tree, so the
// result is bitwise reproducible; a 1 560-chunk hub (RMAT, in-degree ~1e5) is 20 blocks x ~100 dependent
// load rounds instead of ONE block walking 1 560 tiles element by element (round 1: 15.7 ms per call).
template <int VEC>
__global__ __launch_bounds__(256) void hub_combine_kernel(const HubDesc *__restrict__ descs, const float *__restrict__ P,
                                                          ampconv_view_t out, const int32_t *__restrict__ ptr, int L,
                                                          int D, int dh, float scale, int out_bf16,
                                                          const int32_t *__restrict__ header, float *absmax) {
  if ((int)blockIdx.x >= header[2]) return;          // the grid is sized for the most rows n_chunks can hold
  const int64_t c = header[header[3] + blockIdx.x];
  const HubDesc d = descs[c];
  __shared__ float part[4][64 * VEC];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t LD = (int64_t)L * D;
  const int64_t e0 = ((int64_t)blockIdx.y * 64 + lane) * VEC;
  float a0[VEC], a1[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) a0[v] = a1[v] = 0.f;
  if (e0 < LD) {
    const float *p = P + c * LD + e0;
    int k = q;
    for (; k + 4 < d.nfirst; k += 8) {
      if (VEC == 4) {
        const float4 x = *reinterpret_cast<const float4 *>(p + (int64_t)k * LD);
        const float4 y = *reinterpret_cast<const float4 *>(p + (int64_t)(k + 4) * LD);
        a0[0] += x.x; a0[1 % VEC] += x.y; a0[2 % VEC] += x.z; a0[3 % VEC] += x.w;
        a1[0] += y.x; a1[1 % VEC] += y.y; a1[2 % VEC] += y.z; a1[3 % VEC] += y.w;
      } else {
        a0[0] += p[(int64_t)k * LD];
        a1[0] += p[(int64_t)(k + 4) * LD];
      }
    }
    if (k < d.nfirst) {
      if (VEC == 4) {
        const float4 x = *reinterpret_cast<const float4 *>(p + (int64_t)k * LD);
        a0[0] += x.x; a0[1 % VEC] += x.y; a0[2 % VEC] += x.z; a0[3 % VEC] += x.w;
      } else {
        a0[0] += p[(int64_t)k * LD];
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) part[q][lane * VEC + v] = a0[v] + a1[v];
  __syncthreads();
  if (q != 0 || e0 >= LD) return;
  const float sc = ptr ? 1.f / (float)(ptr[d.row + 1] - ptr[d.row]) : scale;   // forward: the mean
  float r[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const int i = lane * VEC + v;
    r[v] = ((part[0][i] + part[1][i]) + (part[2][i] + part[3][i])) * sc;
  }
  const int l = (int)(e0 / D), cc = (int)(e0 - (int64_t)l * D);
  const int64_t off = (int64_t)d.row * out.node_stride + (int64_t)l * out.row_stride +
                      (int64_t)(cc / dh) * out.head_stride + cc % dh;
  if (out_bf16) {
    __bf16 *o = reinterpret_cast<__bf16 *>(out.ptr) + off;
#pragma unroll
    for (int v = 0; v < VEC; ++v) o[v] = (__bf16)r[v];
  } else if (VEC == 4) {
    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(out.ptr) + off) = make_float4(r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]);
  } else {
    reinterpret_cast<float *>(out.ptr)[off] = r[0];
  }
  if (absmax) {            // (lanes of phase 0 that hold elements; the others left above)
    float m = 0.f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) m = fmaxf(m, finite_abs(r[v]));
    if (m > __builtin_nontemporal_load(absmax)) atomicMax(reinterpret_cast<unsigned *>(absmax), __builtin_bit_cast(unsigned, m));
  }
}

}  // namespace

int64_t ampconv_hub_max_chunks(int64_t E, int chunk) { return hub_max_chunks(E, chunk); }

extern "C" size_t ampconv_hub_plan_bytes(int64_t E, int chunk) {
  if (E < 0 || chunk <= 0) return 0;
  const size_t mc = (size_t)hub_max_chunks(E, chunk);
  return 16 + sizeof(HubDesc) * mc + sizeof(int32_t) * (mc / 2 + 2);
}

extern "C" int ampconv_hub_plan(const int32_t *ptr, int64_t N, int64_t E, int chunk, void *plan,
                                void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!ptr || !plan || N <= 0 || E < 0 || chunk <= 0) return AMPCONV_E_BADARG;
  int32_t *header = (int32_t *)plan;
  hipError_t e = hipMemsetAsync(header, 0, 16, stream);
  if (e != hipSuccess) return (int)e;
  HubDesc *descs = (HubDesc *)(header + 4);
  int32_t *firsts = (int32_t *)(descs + hub_max_chunks(E, chunk));
  hub_plan_kernel<<<(unsigned)((N + 255) / 256), 256, 0, stream>>>(ptr, N, chunk, header, descs, firsts);
  return ampconv_launch_status();
}

extern "C" size_t ampconv_hub_workspace_bytes(int64_t n_chunks, int L, int D, int n_tiles) {
  if (n_chunks <= 0 || L <= 0 || D <= 0 || n_tiles <= 0) return 0;
  return (size_t)n_chunks * L * D * sizeof(float) * n_tiles;
}

int ampconv_hub_combine(const void *plan, int64_t n_chunks, const float *P, ampconv_view_t out,
                        const int32_t *ptr_for_mean, int L, int D, int H, float scale, int out_bf16,
                        hipStream_t stream, float *absmax) {
  if (n_chunks <= 0) return AMPCONV_OK;
  const HubDesc *descs = (const HubDesc *)((const int32_t *)plan + 4);
  const int dh = D / H;
  const int64_t LD = (int64_t)L * D;
  // 4 consecutive elements stay inside one head slice and 16-byte aligned in P and in the output
  const bool vec4 = dh % 4 == 0 && out.node_stride % 4 == 0 && out.row_stride % 4 == 0 && out.head_stride % 4 == 0 &&
                    (uintptr_t)P % 16 == 0 && (out_bf16 || (uintptr_t)out.ptr % 16 == 0);
  if (n_chunks > INT32_MAX) return AMPCONV_E_BADARG;
  const int32_t *header = (const int32_t *)plan;
  const unsigned max_rows = (unsigned)(n_chunks / 2 + 1);      // a long row has at least two chunks
  if (vec4) {
    const dim3 grid(max_rows, (unsigned)((LD + 255) / 256));
    hub_combine_kernel<4><<<grid, 256, 0, stream>>>(descs, P, out, ptr_for_mean, L, D, dh, scale, out_bf16, header, absmax);
  } else {
    const dim3 grid(max_rows, (unsigned)((LD + 63) / 64));
    hub_combine_kernel<1><<<grid, 256, 0, stream>>>(descs, P, out, ptr_for_mean, L, D, dh, scale, out_bf16, header, absmax);
  }
  return ampconv_launch_status();
}
