// Probe of v_mfma_f32_4x4x1_16b_f32 on gfx950 (round 2): lane maps of the A / B
// operands and the C/D registers, what blgp / cbsz+abid do to them, what v_permlane16_swap /
// v_permlane32_swap move, and the issue cost of the instruction alone and mixed with
// v_mfma_f32_16x16x4_f32.  Diagnostic tool, not part of libampconv.so.
//   built by __graft_entry__.build() with the product flags (MFMA accumulators in VGPRs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF4(a, b, c, cb, ab, bl) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), cb, ab, bl)
#define MF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int CB, int AB, int BL>
__global__ void map_kernel(float *out, const float *a, const float *b) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = MF4(a[l], b[l], c, CB, AB, BL);
  for (int r = 0; r < 4; ++r) out[4 * l + r] = c[r];
}

__global__ void swap_kernel(unsigned *out) {
  const unsigned l = threadIdx.x;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));   // typed result: see csrc/mfma_tile.h, swap16
  const u32x2 r16 = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
  const u32x2 r32 = __builtin_amdgcn_permlane32_swap(l, 100u + l, false, false);
  out[4 * l + 0] = r16.x; out[4 * l + 1] = r16.y;
  out[4 * l + 2] = r32.x; out[4 * l + 3] = r32.y;
}

// timing: MODE 0: 4x4x1 on 4 independent accumulators; 1: 4x4x1 on one accumulator (dependent);
// 2: 16x16x4 on 4 accumulators; 3: one 16x16x4 then four 4x4x1 (all independent accumulators);
// 4: 4x4x1 with blgp:5 on 4 accumulators; 5: 4x4x1 followed by a dependent v_add chain (VALU beside)
template <int MODE>
__global__ __launch_bounds__(256) void time_kernel(float *sink, unsigned long long *cycles, int iters, float x) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, d0 = c0;
  float a = x + threadIdx.x, b = x - threadIdx.x, v = x;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      c0 = MF4(a, b, c0, 0, 0, 0); c1 = MF4(a, b, c1, 0, 0, 0);
      c2 = MF4(a, b, c2, 0, 0, 0); c3 = MF4(a, b, c3, 0, 0, 0);
    } else if (MODE == 1) {
      c0 = MF4(a, b, c0, 0, 0, 0); c0 = MF4(a, b, c0, 0, 0, 0);
      c0 = MF4(a, b, c0, 0, 0, 0); c0 = MF4(a, b, c0, 0, 0, 0);
    } else if (MODE == 2) {
      c0 = MF16(a, b, c0); c1 = MF16(a, b, c1); c2 = MF16(a, b, c2); c3 = MF16(a, b, c3);
    } else if (MODE == 3) {
      d0 = MF16(a, b, d0);
      c0 = MF4(a, b, c0, 0, 0, 0); c1 = MF4(a, b, c1, 0, 0, 0);
      c2 = MF4(a, b, c2, 0, 0, 0); c3 = MF4(a, b, c3, 0, 0, 0);
    } else if (MODE == 4) {
      c0 = MF4(a, b, c0, 0, 0, 5); c1 = MF4(a, b, c1, 0, 0, 6);
      c2 = MF4(a, b, c2, 0, 0, 7); c3 = MF4(a, b, c3, 0, 0, 4);
    } else if (MODE == 6) {       // phases as in an edge kernel: 8 x 16x16x4 on two accumulators, then 8 x 4x4x1 on two
      for (int j = 0; j < 4; ++j) { d0 = MF16(a, b, d0); c3 = MF16(a, b, c3); }
      for (int j = 0; j < 4; ++j) { c0 = MF4(a, b, c0, 0, 0, 0); c1 = MF4(a, b, c1, 0, 0, 0); }
    } else if (MODE == 7) {       // strict alternation 16x16x4 / 4x4x1
      d0 = MF16(a, b, d0); c0 = MF4(a, b, c0, 0, 0, 0); c3 = MF16(a, b, c3); c1 = MF4(a, b, c1, 0, 0, 0);
    } else {
      c0 = MF4(a, b, c0, 0, 0, 0); v = fmaf(v, 1.0001f, 0.5f);
      c1 = MF4(a, b, c1, 0, 0, 0); v = fmaf(v, 1.0001f, 0.5f);
      c2 = MF4(a, b, c2, 0, 0, 0); v = fmaf(v, 1.0001f, 0.5f);
      c3 = MF4(a, b, c3, 0, 0, 0); v = fmaf(v, 1.0001f, 0.5f);
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  f32x4 s = c0 + c1 + c2 + c3 + d0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + v;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int CB, int AB, int BL>
static void show_map(const char *name) {
  float *da, *db, *dout;
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 1024);
  std::vector<float> ha(64), hb(64), h1(64, 1.f), oa(256), ob(256);
  for (int l = 0; l < 64; ++l) { ha[l] = (float)(l + 1); hb[l] = (float)(l + 1); }
  hipMemcpy(da, ha.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(db, h1.data(), 256, hipMemcpyHostToDevice);
  map_kernel<CB, AB, BL><<<1, 64>>>(dout, da, db);
  hipMemcpy(oa.data(), dout, 1024, hipMemcpyDeviceToHost);
  hipMemcpy(da, h1.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), 256, hipMemcpyHostToDevice);
  map_kernel<CB, AB, BL><<<1, 64>>>(dout, da, db);
  hipMemcpy(ob.data(), dout, 1024, hipMemcpyDeviceToHost);
  printf("== %s: D[lane][reg] = A[lane a] * B[lane b], printed as a/b (source lanes)\n", name);
  for (int l = 0; l < 64; ++l) {
    printf("  lane %2d:", l);
    for (int r = 0; r < 4; ++r) printf(" %2d/%2d", (int)oa[4 * l + r] - 1, (int)ob[4 * l + r] - 1);
    printf("\n");
  }
  hipFree(da); hipFree(db); hipFree(dout);
}

template <int MODE>
static void run_time(const char *name, int blocks_per_cu, int n_mfma_per_iter) {
  const int iters = 4096, blocks = 256 * blocks_per_cu;
  float *sink; unsigned long long *cyc;
  hipMalloc(&sink, sizeof(float) * blocks * 256);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4);
  time_kernel<MODE><<<blocks, 256>>>(sink, cyc, iters, 1.0f);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  time_kernel<MODE><<<blocks, 256>>>(sink, cyc, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0, mn = 1e30, mx = 0;
  for (auto v : h) { sum += (double)v; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
  const double den = (double)iters * n_mfma_per_iter, per = sum / h.size() / den;
  printf("%-44s waves/SIMD %d: s_memtime ticks per MFMA per wave avg %.2f min %.2f max %.2f; kernel %.3f ms = %.2f ns "
         "per MFMA per SIMD\n", name, blocks_per_cu, per, mn / den, mx / den, ms, ms * 1e6 / (den * blocks_per_cu));
  hipFree(sink); hipFree(cyc);
}

int main() {
  show_map<0, 0, 0>("4x4x1 plain");
  show_map<0, 0, 4>("4x4x1 blgp:4");
  show_map<0, 0, 5>("4x4x1 blgp:5");
  show_map<0, 0, 1>("4x4x1 blgp:1");
  show_map<0, 0, 3>("4x4x1 blgp:3");
  show_map<4, 3, 0>("4x4x1 cbsz:4 abid:3");
  show_map<2, 1, 0>("4x4x1 cbsz:2 abid:1");
  {
    unsigned *d; hipMalloc(&d, 1024);
    swap_kernel<<<1, 64>>>(d);
    std::vector<unsigned> h(256); hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    printf("== permlane16_swap(v0 = lane, v1 = 100 + lane) -> (v0', v1') ; permlane32_swap likewise\n");
    for (int l = 0; l < 64; ++l) printf("  lane %2d: p16 %3u %3u   p32 %3u %3u\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    hipFree(d);
  }
  for (int w = 1; w <= 4; ++w) {
    if (w == 3) continue;
    run_time<0>("4x4x1, 4 independent accumulators", w, 4);
    run_time<1>("4x4x1, one accumulator (dependent)", w, 4);
    run_time<2>("16x16x4, 4 independent accumulators", w, 4);
    run_time<3>("1 x 16x16x4 + 4 x 4x4x1 (per 5 MFMAs)", w, 5);
    run_time<4>("4x4x1 blgp 4..7, 4 accumulators", w, 4);
    run_time<5>("4x4x1 + v_fma chain (per MFMA)", w, 4);
    run_time<6>("8 x 16x16x4 then 8 x 4x4x1 (per 16 MFMAs)", w, 16);
    run_time<7>("alternating 16x16x4 / 4x4x1 (per 4 MFMAs)", w, 4);
  }
  return 0;
}
