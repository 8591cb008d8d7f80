// Micro-benchmark: how fast can one wavefront per (destination, head) gather the
// 20 x dh head tiles of random source nodes, as a function of the HBM layout of
// K/V?  Decides the layout contract between the projection GEMMs and the edge
// kernels (DESIGN.md "Data layout").  Stand-alone: hipcc tools/gather_bench.hip.
//
//   rowmajor : [N][L][D]      tile rows are 128-B segments at a 1-KB stride
//   packed3  : [N][L][3D]     same, 3-KB stride (Q|K|V written by one GEMM)
//   headmajor: [N][H][L][dh]  tile is one contiguous 2.5-KB block
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int L = 20, DH = 32, H = 8, D = 256;

// block = 8 waves = the 8 heads of one destination; deg sources per destination
__global__ __launch_bounds__(512) void gather_tiles(const float *__restrict__ K, const float *__restrict__ V,
                                                    const int *__restrict__ col, int deg,
                                                    long node_stride, long row_stride, long head_stride,
                                                    float *__restrict__ out) {
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const long d = blockIdx.x;
  const int r = lane >> 3, c4 = (lane & 7) * 4;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int p = 0; p < deg; ++p) {
    const long s = col[d * deg + p];
    const float *k = K + s * node_stride + h * head_stride + c4;
    const float *v = V + s * node_stride + h * head_stride + c4;
    float4 k0 = *(const float4 *)(k + (long)r * row_stride);
    float4 k1 = *(const float4 *)(k + (long)(r + 8) * row_stride);
    float4 v0 = *(const float4 *)(v + (long)r * row_stride);
    float4 v1 = *(const float4 *)(v + (long)(r + 8) * row_stride);
    float4 k2 = make_float4(0, 0, 0, 0), v2 = k2;
    if (r < 4) {
      k2 = *(const float4 *)(k + (long)(r + 16) * row_stride);
      v2 = *(const float4 *)(v + (long)(r + 16) * row_stride);
    }
    acc.x += k0.x + k1.x + k2.x + v0.x + v1.x + v2.x;
    acc.y += k0.y + k1.y + k2.y + v0.y + v1.y + v2.y;
    acc.z += k0.z + k1.z + k2.z + v0.z + v1.z + v2.z;
    acc.w += k0.w + k1.w + k2.w + v0.w + v1.w + v2.w;
  }
  out[(d * H + h) * 64 + lane] = acc.x + acc.y + acc.z + acc.w;
}

// whole 20-KB rows by one wave (the guide's measured pattern), for reference
__global__ __launch_bounds__(256) void gather_rows(const float *__restrict__ K, const int *__restrict__ col,
                                                   int deg, long row_floats, float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long d = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  float4 acc = make_float4(0, 0, 0, 0);
  for (int p = 0; p < deg; ++p) {
    const float *k = K + (long)col[d * deg + p] * row_floats;
    for (int o = lane * 4; o < row_floats; o += 256) {
      float4 t = *(const float4 *)(k + o);
      acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
  }
  out[d * 64 + lane] = acc.x + acc.y + acc.z + acc.w;
}

int main(int argc, char **argv) {
  const long N = argc > 1 ? atol(argv[1]) : 200000;       // 200k nodes x 20 KB = 4 GB per table
  const long ND = argc > 2 ? atol(argv[2]) : 100000;      // destinations
  const int deg = 10;
  float *K, *V, *out; int *col;
  const size_t tbl = (size_t)N * L * 3 * D * sizeof(float);   // big enough for packed3
  CK(hipMalloc(&K, tbl));
  CK(hipMalloc(&V, (size_t)N * L * D * sizeof(float)));
  CK(hipMalloc(&out, (size_t)ND * H * 64 * sizeof(float)));
  CK(hipMalloc(&col, (size_t)ND * deg * sizeof(int)));
  CK(hipMemset(K, 0, tbl));
  CK(hipMemset(V, 0, (size_t)N * L * D * sizeof(float)));
  std::vector<int> hc((size_t)ND * deg);
  unsigned long long st = 88172645463325252ull;
  for (auto &x : hc) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; x = (int)(st % (unsigned long long)N); }
  CK(hipMemcpy(col, hc.data(), hc.size() * sizeof(int), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Cfg { const char *name; long node, row, head; const float *k, *v; };
  Cfg cfgs[] = {
    {"rowmajor  [N][L][D]     ", (long)L * D, D, DH, K, V},
    {"packed3   [N][L][3D]    ", (long)L * 3 * D, 3 * D, DH, K + D, K + 2 * D},
    {"headmajor [N][H][L][dh] ", (long)L * D, DH, (long)L * DH, K, V},
  };
  const double bytes = (double)ND * deg * 2.0 * L * D * sizeof(float);
  for (auto &c : cfgs) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      gather_tiles<<<ND, 512>>>(c.k, c.v, col, deg, c.node, c.row, c.head, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("%s tiles: %.3f ms  %.1f GB/s\n", c.name, ms, bytes / ms * 1e-6);
    }
  }
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    gather_rows<<<ND / 4, 256>>>(K, col, deg, (long)L * D, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 2) printf("whole 20-KB rows, wave per destination: %.3f ms  %.1f GB/s\n", ms, bytes / 2 / ms * 1e-6);
  }
  CK(hipGetLastError());
  return 0;
}
