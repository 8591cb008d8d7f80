// Probe (gfx950): does a COUNTED `s_waitcnt lgkmcnt(N)`, N > 0, cover the ds_read_b64_tr_b16 reads older than the N
// youngest -- as it does for ordinary LDS reads, which return in order?
//
// Why: the weight-gradient kernel (csrc/proj_gemm.hip) gave wrong sums in 7-100 % of its launches in ONE schedule and
// never in another.  Side by side, the two instruction streams differ in this: the failing one consumes
// its transposed fragments behind counted waits (`... ds_read_b64_tr_b16 x 10; s_waitcnt lgkmcnt(6); v_mfma;
// s_waitcnt lgkmcnt(2); v_mfma ...`), the passing one only behind `s_waitcnt lgkmcnt(0)`.  EXEC is all ones at every
// transposed read in both, the addresses are the same expressions (8-byte aligned), no global load is issued between a
// read and its consumer in either (tools/scan_tr_hazard.py --diff).  The wrong values were always one 16-bit element in
// lanes 16-31 / 48-63: the second 16-lane group of each 32-lane half of the transposing read.  The round-4 probe
// (tools/tr_hazard_probe.hip) only ever waited with lgkmcnt(0) and saw nothing.
//
// What it does: per iteration a wave pre-fills 16 destination registers with a sentinel, issues 8 reads (transposing, or
// plain ds_read_b64 as the control) of a known image, waits `lgkmcnt(6)`, copies the registers of reads 0-1 (v_mov: the
// consumer), waits `lgkmcnt(2)`, copies reads 2-5, then drains (`lgkmcnt(0)` + 16 idle cycles) and compares the copies
// with the registers as they stand after the drain.  A copy that still holds the sentinel (or anything else) = the
// counted wait let the consumer run before the data had landed.  Other waves of the CU keep the LDS and the memory
// pipe busy (same code, staggered), as in the kernel.
//
//   hipcc --offload-arch=gfx950 -O2 tools/tr_lgkm_probe.hip -o tools/tr_lgkm_probe && tools/tr_lgkm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool TR>
__global__ __launch_bounds__(256) void probe(const unsigned *__restrict__ fill, unsigned long long *__restrict__ bad, int iters,
                                             const float *__restrict__ stream, float *__restrict__ sink) {
  __shared__ __attribute__((aligned(16))) unsigned img[8 * 1024 / 4 * 2];      // 16 KiB
  for (int i = threadIdx.x; i < 4096; i += 256) img[i] = fill[i] * 2654435761u + i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // the address pattern of the kernel's fragment reads: lane 4 q + p of a 16-lane group -> row q, columns 4 p .. 4 p + 3
  const unsigned a0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned *)img + ((lane >> 4) * 2048) +
                      (((lane >> 2) & 3) * 256) + ((lane & 3) * 8);
  unsigned long long nbad = 0;
  float acc = 0.f;
  const float *sp = stream + (size_t)blockIdx.x * 65536 + threadIdx.x * 4;
  for (int it = 0; it < iters; ++it) {
    // a global load in flight beside the LDS reads, as in the kernel's stage loop
    const float4 g = *reinterpret_cast<const float4 *>(sp + (size_t)(it & 63) * 1024);
    unsigned d;
    if (TR) {
      asm volatile(
          "v_mov_b32 v100, 0xdead\n\tv_mov_b32 v101, 0xdead\n\tv_mov_b32 v102, 0xdead\n\tv_mov_b32 v103, 0xdead\n\t"
          "v_mov_b32 v104, 0xdead\n\tv_mov_b32 v105, 0xdead\n\tv_mov_b32 v106, 0xdead\n\tv_mov_b32 v107, 0xdead\n\t"
          "v_mov_b32 v108, 0xdead\n\tv_mov_b32 v109, 0xdead\n\tv_mov_b32 v110, 0xdead\n\tv_mov_b32 v111, 0xdead\n\t"
          "s_nop 4\n\t"
          "ds_read_b64_tr_b16 v[100:101], %1\n\tds_read_b64_tr_b16 v[102:103], %1 offset:64\n\t"
          "ds_read_b64_tr_b16 v[104:105], %1 offset:1024\n\tds_read_b64_tr_b16 v[106:107], %1 offset:1088\n\t"
          "ds_read_b64_tr_b16 v[108:109], %1 offset:128\n\tds_read_b64_tr_b16 v[110:111], %1 offset:192\n\t"
          "ds_read_b64_tr_b16 v[112:113], %1 offset:1152\n\tds_read_b64_tr_b16 v[114:115], %1 offset:1216\n\t"
          "s_waitcnt lgkmcnt(6)\n\t"
          "v_mov_b32 v120, v100\n\tv_mov_b32 v121, v101\n\tv_mov_b32 v122, v102\n\tv_mov_b32 v123, v103\n\t"
          "s_waitcnt lgkmcnt(2)\n\t"
          "v_mov_b32 v124, v104\n\tv_mov_b32 v125, v105\n\tv_mov_b32 v126, v106\n\tv_mov_b32 v127, v107\n\t"
          "v_mov_b32 v128, v108\n\tv_mov_b32 v129, v109\n\tv_mov_b32 v130, v110\n\tv_mov_b32 v131, v111\n\t"
          "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\t"
          "v_xor_b32 v120, v120, v100\n\tv_xor_b32 v121, v121, v101\n\tv_xor_b32 v122, v122, v102\n\tv_xor_b32 v123, v123, v103\n\t"
          "v_xor_b32 v124, v124, v104\n\tv_xor_b32 v125, v125, v105\n\tv_xor_b32 v126, v126, v106\n\tv_xor_b32 v127, v127, v107\n\t"
          "v_xor_b32 v128, v128, v108\n\tv_xor_b32 v129, v129, v109\n\tv_xor_b32 v130, v130, v110\n\tv_xor_b32 v131, v131, v111\n\t"
          "v_or3_b32 v120, v120, v121, v122\n\tv_or3_b32 v123, v123, v124, v125\n\tv_or3_b32 v126, v126, v127, v128\n\t"
          "v_or3_b32 v129, v129, v130, v131\n\tv_or3_b32 v120, v120, v123, v126\n\tv_or_b32 %0, v120, v129\n\t"
          : "=v"(d)
          : "v"(a0)
          : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",
            "v113", "v114", "v115", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131");
    } else {
      asm volatile(
          "v_mov_b32 v100, 0xdead\n\tv_mov_b32 v101, 0xdead\n\tv_mov_b32 v102, 0xdead\n\tv_mov_b32 v103, 0xdead\n\t"
          "v_mov_b32 v104, 0xdead\n\tv_mov_b32 v105, 0xdead\n\tv_mov_b32 v106, 0xdead\n\tv_mov_b32 v107, 0xdead\n\t"
          "v_mov_b32 v108, 0xdead\n\tv_mov_b32 v109, 0xdead\n\tv_mov_b32 v110, 0xdead\n\tv_mov_b32 v111, 0xdead\n\t"
          "s_nop 4\n\t"
          "ds_read_b64 v[100:101], %1\n\tds_read_b64 v[102:103], %1 offset:64\n\t"
          "ds_read_b64 v[104:105], %1 offset:1024\n\tds_read_b64 v[106:107], %1 offset:1088\n\t"
          "ds_read_b64 v[108:109], %1 offset:128\n\tds_read_b64 v[110:111], %1 offset:192\n\t"
          "ds_read_b64 v[112:113], %1 offset:1152\n\tds_read_b64 v[114:115], %1 offset:1216\n\t"
          "s_waitcnt lgkmcnt(6)\n\t"
          "v_mov_b32 v120, v100\n\tv_mov_b32 v121, v101\n\tv_mov_b32 v122, v102\n\tv_mov_b32 v123, v103\n\t"
          "s_waitcnt lgkmcnt(2)\n\t"
          "v_mov_b32 v124, v104\n\tv_mov_b32 v125, v105\n\tv_mov_b32 v126, v106\n\tv_mov_b32 v127, v107\n\t"
          "v_mov_b32 v128, v108\n\tv_mov_b32 v129, v109\n\tv_mov_b32 v130, v110\n\tv_mov_b32 v131, v111\n\t"
          "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\t"
          "v_xor_b32 v120, v120, v100\n\tv_xor_b32 v121, v121, v101\n\tv_xor_b32 v122, v122, v102\n\tv_xor_b32 v123, v123, v103\n\t"
          "v_xor_b32 v124, v124, v104\n\tv_xor_b32 v125, v125, v105\n\tv_xor_b32 v126, v126, v106\n\tv_xor_b32 v127, v127, v107\n\t"
          "v_xor_b32 v128, v128, v108\n\tv_xor_b32 v129, v129, v109\n\tv_xor_b32 v130, v130, v110\n\tv_xor_b32 v131, v131, v111\n\t"
          "v_or3_b32 v120, v120, v121, v122\n\tv_or3_b32 v123, v123, v124, v125\n\tv_or3_b32 v126, v126, v127, v128\n\t"
          "v_or3_b32 v129, v129, v130, v131\n\tv_or3_b32 v120, v120, v123, v126\n\tv_or_b32 %0, v120, v129\n\t"
          : "=v"(d)
          : "v"(a0)
          : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",
            "v113", "v114", "v115", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131");
    }
    nbad += d != 0;
    acc += g.x + g.y + g.z + g.w;
  }
  if (acc == 12345.678f) sink[0] = acc;                  // keeps the global loads
  // per 16-lane group of the wave: which groups saw a stale copy
  atomicAdd(&bad[(lane >> 4)], nbad);
}

// The same question with the kernel's own consumer: an MFMA right behind the counted wait.  MFMA 1 reads the two
// fragments of reads 0-3 behind `lgkmcnt(4)`, MFMA 2 reads the same registers after the drain; their 16 accumulators must
// agree bit for bit.  (A vector instruction reads its sources a lane quarter at a time; the matrix instruction fetches
// SrcA / SrcB differently, so the v_mov probe above does not cover it.)
template <bool TR>
__global__ __launch_bounds__(256) void probe_mfma(const unsigned *__restrict__ fill, unsigned long long *__restrict__ bad,
                                                  int iters, const float *__restrict__ stream, float *__restrict__ sink) {
  __shared__ __attribute__((aligned(16))) unsigned img[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) img[i] = fill[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const unsigned a0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned *)img + ((lane >> 4) * 2048) +
                      (((lane >> 2) & 3) * 256) + ((lane & 3) * 8);
  unsigned long long nbad = 0;
  float acc = 0.f;
  const float *sp = stream + (size_t)blockIdx.x * 65536 + threadIdx.x * 4;
  for (int it = 0; it < iters; ++it) {
    const float4 g = *reinterpret_cast<const float4 *>(sp + (size_t)(it & 63) * 1024);
    unsigned d;
#define PROBE_BODY(RD)                                                                                                       \
      "v_mov_b32 v100, 0x40004000\n\tv_mov_b32 v101, 0x40004000\n\tv_mov_b32 v102, 0x40004000\n\tv_mov_b32 v103, 0x40004000\n\t" \
      "v_mov_b32 v104, 0x40004000\n\tv_mov_b32 v105, 0x40004000\n\tv_mov_b32 v106, 0x40004000\n\tv_mov_b32 v107, 0x40004000\n\t" \
      "s_nop 4\n\t"                                                                                                          \
      RD " v[100:101], %1\n\t" RD " v[102:103], %1 offset:64\n\t"                                                            \
      RD " v[104:105], %1 offset:1024\n\t" RD " v[106:107], %1 offset:1088\n\t"                                              \
      RD " v[108:109], %1 offset:128\n\t" RD " v[110:111], %1 offset:192\n\t"                                                \
      RD " v[112:113], %1 offset:1152\n\t" RD " v[114:115], %1 offset:1216\n\t"                                              \
      "s_waitcnt lgkmcnt(4)\n\t"                                                                                             \
      "v_mfma_f32_32x32x16_bf16 v[140:155], v[100:103], v[104:107], 0\n\t"                                                   \
      "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\t"                                                                      \
      "v_mfma_f32_32x32x16_bf16 v[160:175], v[100:103], v[104:107], 0\n\t"                                                   \
      "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"                                                                                  \
      "v_xor_b32 v140, v140, v160\n\tv_xor_b32 v141, v141, v161\n\tv_xor_b32 v142, v142, v162\n\tv_xor_b32 v143, v143, v163\n\t" \
      "v_xor_b32 v144, v144, v164\n\tv_xor_b32 v145, v145, v165\n\tv_xor_b32 v146, v146, v166\n\tv_xor_b32 v147, v147, v167\n\t" \
      "v_xor_b32 v148, v148, v168\n\tv_xor_b32 v149, v149, v169\n\tv_xor_b32 v150, v150, v170\n\tv_xor_b32 v151, v151, v171\n\t" \
      "v_xor_b32 v152, v152, v172\n\tv_xor_b32 v153, v153, v173\n\tv_xor_b32 v154, v154, v174\n\tv_xor_b32 v155, v155, v175\n\t" \
      "v_or3_b32 v140, v140, v141, v142\n\tv_or3_b32 v143, v143, v144, v145\n\tv_or3_b32 v146, v146, v147, v148\n\t"          \
      "v_or3_b32 v149, v149, v150, v151\n\tv_or3_b32 v152, v152, v153, v154\n\tv_or3_b32 v140, v140, v143, v146\n\t"          \
      "v_or3_b32 %0, v140, v149, v152\n\tv_or_b32 %0, %0, v155\n\t"
#define PROBE_CLOBBER                                                                                                        \
  "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113",  \
      "v114", "v115", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", \
      "v153", "v154", "v155", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", \
      "v172", "v173", "v174", "v175"
    if (TR) asm volatile(PROBE_BODY("ds_read_b64_tr_b16") : "=&v"(d) : "v"(a0) : PROBE_CLOBBER);
    else asm volatile(PROBE_BODY("ds_read_b64") : "=&v"(d) : "v"(a0) : PROBE_CLOBBER);
    nbad += d != 0;
    acc += g.x + g.y + g.z + g.w;
  }
  if (acc == 12345.678f) sink[0] = acc;
  atomicAdd(&bad[(lane >> 4)], nbad);
}

int main() {
  const int blocks = 2048, iters = 20000;
  unsigned *fill;
  unsigned long long *bad;
  float *stream, *sink;
  CHECK(hipMalloc(&fill, 4096 * 4));
  CHECK(hipMalloc(&bad, 8 * 8));
  CHECK(hipMalloc(&stream, (size_t)blocks * 65536 * 4 + 1024 * 64 * 4));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(fill, 0x5a, 4096 * 4));
  CHECK(hipMemset(stream, 0, (size_t)blocks * 65536 * 4 + 1024 * 64 * 4));
  for (int tr = 0; tr < 2; ++tr) {
    CHECK(hipMemset(bad, 0, 64));
    if (tr) probe<true><<<blocks, 256>>>(fill, bad, iters, stream, sink);
    else probe<false><<<blocks, 256>>>(fill, bad, iters, stream, sink);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[8];
    CHECK(hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost));
    const double total = (double)blocks * 256 * iters;
    printf("%s: lanes with a stale copy behind a counted lgkmcnt wait, by 16-lane group of the wave: %llu %llu %llu %llu of %.3g "
           "lane-iterations each\n", tr ? "ds_read_b64_tr_b16" : "ds_read_b64 (control)", h[0], h[1], h[2], h[3], total / 4);
  }
  // image of small integers as bf16 pairs: every sum is exact, no NaN patterns
  {
    unsigned *hf = (unsigned *)malloc(4096 * 4);
    for (int i = 0; i < 4096; ++i) {
      const float lo = (float)((i * 7) & 15), hi = (float)((i * 13 + 5) & 15);
      unsigned ul, uh;
      memcpy(&ul, &lo, 4);
      memcpy(&uh, &hi, 4);
      hf[i] = (ul >> 16) | (uh & 0xFFFF0000u);
    }
    CHECK(hipMemcpy(fill, hf, 4096 * 4, hipMemcpyHostToDevice));
    free(hf);
  }
  for (int tr = 0; tr < 2; ++tr) {
    CHECK(hipMemset(bad, 0, 64));
    if (tr) probe_mfma<true><<<blocks, 256>>>(fill, bad, iters, stream, sink);
    else probe_mfma<false><<<blocks, 256>>>(fill, bad, iters, stream, sink);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[8];
    CHECK(hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost));
    const double total = (double)blocks * 256 * iters;
    printf("%s + MFMA consumer: lanes whose accumulators differ between the MFMA behind lgkmcnt(4) and the one behind the "
           "drain, by 16-lane group: %llu %llu %llu %llu of %.3g lane-iterations each\n",
           tr ? "ds_read_b64_tr_b16" : "ds_read_b64 (control)", h[0], h[1], h[2], h[3], total / 4);
  }
  return 0;
}
