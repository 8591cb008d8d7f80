// Probe (gfx950): is it safe to redefine an MFMA's SrcA registers with ds_read_b64_tr_b16 right behind the MFMA, and to
// feed the freshly read registers to the next MFMA right behind the wait?  (The schedule hipcc picked for the fp32
// weight-gradient kernel, csrc/proj_gemm.hip, which gave wrong sums in 7-100 % of the launches: DESIGN.md 4a.)
//
// One wave sequence, explicit registers:
//     ds_read_b64_tr_b16 v[200:201] <- image X0 ; ds_read_b64_tr_b16 v[202:203] <- X0 ; s_waitcnt lgkmcnt(0)
//     [four filler MFMAs on other accumulators: the matrix pipe is busy when the MFMA of interest arrives]
//     v_mfma_f32_32x32x16_bf16 v[204:219] = A v[200:203] x B          (result must be X0 . B)
//     s_nop K1
//     ds_read_b64_tr_b16 v[200:201] <- image X1 ; ds_read_b64_tr_b16 v[202:203] <- X1     (redefine SrcA)
//     s_waitcnt lgkmcnt(0) ; s_nop K2
//     v_mfma_f32_32x32x16_bf16 v[220:235] = A v[200:203] x B          (result must be X1 . B)
// X0 holds 1.0 everywhere, X1 holds 2.0, B holds 1.0: every element of the first result must be 16, of the second 32.
// A first result that is not 16 = the redefinition reached the running MFMA (write-after-read); a second result that
// is not 32 = the consumer read its operand before the transposed read had fully landed (read-after-write).
// Two waves per SIMD (240 registers), all running the same sequence: LDS and matrix pipe are contended.  Diagnostic tool, not
// part of libampconv.so.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K1, int K2>
__global__ __launch_bounds__(512) void probe(unsigned *bad0, unsigned *bad1, int iters, const float *stream, size_t nstream,
                                             unsigned *bad2) {
  __shared__ __attribute__((aligned(16))) unsigned short img[2][32 * 64];      // two [32 rows][64 columns] bf16 images
  for (int i = threadIdx.x; i < 32 * 64; i += blockDim.x) {
    img[0][i] = 0x3F80;      // 1.0
    img[1][i] = 0x4000;      // 2.0
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // any valid transposing-read address pattern: 16 lanes cover 4 rows x 16 columns
  const int row = (lane >> 2) & 3, col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3), half = lane >> 5;
  const unsigned a0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)(&img[0][0]) + ((8 * half + row) * 64 + col) * 2;
  const unsigned a1 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)(&img[1][0]) + ((8 * half + row) * 64 + col) * 2;
  unsigned w0 = 0, w1 = 0, w2 = 0;
  // rows streamed from HBM beside the sequence (the kernel's two stages of rows in flight): every element holds 5.0
  const float *gp = stream + ((size_t)blockIdx.x * 512 + threadIdx.x) * 4;
  const size_t gstep = (size_t)gridDim.x * 512 * 4;
  for (int it = 0; it < iters; ++it) {
    float r0, r1, g0, g1, g2;
    const float *p0 = gp + (size_t)(3 * it) * gstep % nstream, *p1 = gp + (size_t)(3 * it + 1) * gstep % nstream,
                *p2 = gp + (size_t)(3 * it + 2) * gstep % nstream;
    asm volatile(
        "global_load_dwordx4 v[188:191], %9, off\n\tglobal_load_dword v196, %10, off\n\tglobal_load_dwordx4 v[192:195], %11, off\n\t"
        "v_mov_b32 v236, 0x3f803f80\n\tv_mov_b32 v237, 0x3f803f80\n\tv_mov_b32 v238, 0x3f803f80\n\tv_mov_b32 v239, 0x3f803f80\n\t"
        "ds_read_b64_tr_b16 v[200:201], %5\n\tds_read_b64_tr_b16 v[202:203], %5 offset:512\n\t"
        "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\t"
        "v_mfma_f32_32x32x16_bf16 v[140:155], v[236:239], v[236:239], 0\n\t"
        "v_mfma_f32_32x32x16_bf16 v[156:171], v[236:239], v[236:239], 0\n\t"
        "v_mfma_f32_32x32x16_bf16 v[172:187], v[236:239], v[236:239], 0\n\t"
        "v_mfma_f32_32x32x16_bf16 v[140:155], v[236:239], v[236:239], 0\n\t"
        "v_mfma_f32_32x32x16_bf16 v[204:219], v[200:203], v[236:239], 0\n\t"
        "s_nop %7\n\t"
        "ds_read_b64_tr_b16 v[200:201], %6\n\tds_read_b64_tr_b16 v[202:203], %6 offset:512\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_nop %8\n\t"
        "v_mfma_f32_32x32x16_bf16 v[220:235], v[200:203], v[236:239], 0\n\t"
        "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_mov_b32 %0, v204\n\tv_mov_b32 %1, v220\n\tv_mov_b32 %2, v188\n\tv_mov_b32 %3, v196\n\tv_mov_b32 %4, v192\n\t"
        : "=v"(r0), "=v"(r1), "=v"(g0), "=v"(g1), "=v"(g2)
        : "v"(a0), "v"(a1), "n"(K1), "n"(K2), "v"(p0), "v"(p1), "v"(p2)
        : "memory", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v140", "v141", "v142", "v143",
          "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152",
          "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166",
          "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180",
          "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v200", "v201", "v202", "v203", "v204", "v205", "v206",
          "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220",
          "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234",
          "v235", "v236", "v237", "v238", "v239");
    w2 += (g0 != 5.f) + (g1 != 5.f) + (g2 != 5.f);
    w0 += r0 != 16.f;
    w1 += r1 != 32.f;
  }
  if (w0) atomicAdd(bad0, w0);
  if (w1) atomicAdd(bad1, w1);
  if (w2) atomicAdd(bad2, w2);
}

template <int K1, int K2>
void run(unsigned *d, int iters, const float *stream, size_t nstream) {
  (void)hipMemset(d, 0, 12);
  probe<K1, K2><<<1024, 512>>>(d, d + 1, iters, stream, nstream, d + 2);
  unsigned h[3];
  (void)hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
  const double n = 1024.0 * 512 * iters;
  printf("nops behind the MFMA %2d, behind the wait %2d: first result wrong %9u (%.2e), second result wrong %9u (%.2e), streamed values wrong %u of %.3g lane results\n",
         K1 + 1, K2 + 1, h[0], h[0] / n, h[1], h[1] / n, h[2], n);
}

int main() {
  unsigned *d;
  (void)hipMalloc(&d, 12);
  const int iters = 2000;
  const size_t nstream = (size_t)1 << 30;                 // 4 GiB of 5.0: the streamed loads miss every cache
  float *stream;
  (void)hipMalloc(&stream, nstream * sizeof(float));
  {
    std::vector<float> five(1 << 20, 5.f);
    for (size_t o = 0; o < nstream; o += five.size()) (void)hipMemcpy(stream + o, five.data(), five.size() * sizeof(float), hipMemcpyHostToDevice);
  }
  run<0, 0>(d, iters, stream, nstream);
  run<3, 0>(d, iters, stream, nstream);
  run<7, 0>(d, iters, stream, nstream);
  run<15, 0>(d, iters, stream, nstream);
  run<0, 3>(d, iters, stream, nstream);
  run<0, 7>(d, iters, stream, nstream);
  run<0, 15>(d, iters, stream, nstream);
  run<15, 15>(d, iters, stream, nstream);
  hipError_t e = hipDeviceSynchronize();
  printf("%s\n", hipGetErrorString(e));
  return e != hipSuccess;
}
