// Long-segment ("hub") plan and the ordered combine of partial tiles (include/ampconv.h,
// "long segments").  Replaces nothing in the reference: PyG's scatter handles any degree; here
// it keeps one-wave-per-(row, head) kernels from running as long as their longest segment
// (BASELINE config 5: RMAT, in-degrees up to ~1e5).
#include "common.h"

namespace {

__global__ void hub_plan_kernel(const int32_t *__restrict__ ptr, int64_t N, int chunk,
                                int32_t *__restrict__ header, HubDesc *__restrict__ descs) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r == 0) header[1] = chunk;
  if (r >= N) return;
  const int beg = ptr[r], end = ptr[r + 1];
  const int deg = end - beg;
  if (deg <= chunk) return;
  const int nc = (deg + chunk - 1) / chunk;
  const int base = atomicAdd(header, nc);      // slot order is arbitrary, results do not depend on it
  for (int k = 0; k < nc; ++k) {
    const int b = beg + k * chunk;
    descs[base + k] = HubDesc{(int32_t)r, b, b + chunk < end ? b + chunk : end, k == 0 ? nc : 0};
  }
}

// out[row] = scale * sum_{k < n} P[c + k]   (chunk order), one block per first chunk
__global__ void hub_combine_kernel(const HubDesc *__restrict__ descs, const float *__restrict__ P,
                                   ampconv_view_t out, const int32_t *__restrict__ ptr, int L, int D,
                                   int dh, float scale, int out_bf16) {
  const int64_t c = blockIdx.x;
  const HubDesc d = descs[c];
  if (d.nfirst == 0) return;
  const int64_t LD = (int64_t)L * D;
  const float sc = ptr ? 1.f / (float)(ptr[d.row + 1] - ptr[d.row]) : scale;   // forward: the mean
  const int64_t base = (int64_t)d.row * out.node_stride;
  for (int64_t e = threadIdx.x; e < LD; e += blockDim.x) {
    float acc = 0.f;
    for (int k = 0; k < d.nfirst; ++k) acc += P[(c + k) * LD + e];
    const int l = (int)(e / D), cc = (int)(e - (int64_t)l * D);
    const int64_t off = base + (int64_t)l * out.row_stride + (int64_t)(cc / dh) * out.head_stride + cc % dh;
    if (out_bf16)
      reinterpret_cast<__bf16 *>(out.ptr)[off] = (__bf16)(acc * sc);
    else
      reinterpret_cast<float *>(out.ptr)[off] = acc * sc;
  }
}

}  // namespace

extern "C" size_t ampconv_hub_plan_bytes(int64_t E, int chunk) {
  if (E < 0 || chunk <= 0) return 0;
  return 16 + sizeof(HubDesc) * (size_t)(2 * (E / chunk) + 2);
}

extern "C" int ampconv_hub_plan(const int32_t *ptr, int64_t N, int64_t E, int chunk, void *plan,
                                void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!ptr || !plan || N <= 0 || E < 0 || chunk <= 0) return AMPCONV_E_BADARG;
  int32_t *header = (int32_t *)plan;
  hipError_t e = hipMemsetAsync(header, 0, 16, stream);
  if (e != hipSuccess) return (int)e;
  hub_plan_kernel<<<(unsigned)((N + 255) / 256), 256, 0, stream>>>(ptr, N, chunk, header,
                                                                  (HubDesc *)(header + 4));
  return ampconv_launch_status();
}

extern "C" size_t ampconv_hub_workspace_bytes(int64_t n_chunks, int L, int D, int n_tiles) {
  if (n_chunks <= 0 || L <= 0 || D <= 0 || n_tiles <= 0) return 0;
  return (size_t)n_chunks * L * D * sizeof(float) * n_tiles;
}

int ampconv_hub_combine(const void *plan, int64_t n_chunks, const float *P, ampconv_view_t out,
                        const int32_t *ptr_for_mean, int L, int D, int H, float scale, int out_bf16,
                        hipStream_t stream) {
  if (n_chunks <= 0) return AMPCONV_OK;
  const HubDesc *descs = (const HubDesc *)((const int32_t *)plan + 4);
  hub_combine_kernel<<<(unsigned)n_chunks, 256, 0, stream>>>(descs, P, out, ptr_for_mean, L, D, D / H,
                                                            scale, out_bf16);
  return ampconv_launch_status();
}
