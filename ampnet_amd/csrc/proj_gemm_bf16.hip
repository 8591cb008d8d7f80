// Node phase of AMPConv with bf16 storage (BASELINE config 5), gfx950.
//
// What it replaces: the packed in-projection and the out-projection of nn.MultiheadAttention
// (torch functional.py:5785-5862 `_in_projection_packed`, :6600 `linear(attn_output, out_proj...)`) and their
// autograd backward, once per NODE, for a layer held in bf16 (`layer.to(torch.bfloat16)`):
// bf16 rows and weights in HBM, ONE v_mfma_f32_32x32x16_bf16 per fragment pair, fp32 accumulate, one rounding
// to bf16 on the way out.  Until round 4 this mode ran on rocBLAS bf16 GEMMs plus torch reductions.
//
// Regime: at K = 256 these products are HBM-bound (qkv: 512 B in, 1536 B out per row = 86 GB at cfg5 against
// 16.5 TFLOP), so both kernels are built around the memory path: EVERY operand byte goes HBM/L2 -> LDS by
// LDS-DMA (global_load_lds_dwordx4, no staging registers, no ds_write), rows as whole 128-byte (or longer)
// lines, the LDS images XOR-swizzled on the SOURCE address so that the linear DMA destination is read back
// conflict-free, and the waits are counted by hand (the counts are compile-time constants: every DMA is issued
// unconditionally from a clamped address).
//
//   proj_rows_bf16   out[M, N] = A[M, K] W^T (+ bias) (* row mask)
//   proj_wgrad_bf16  dW[Na, Nb] = A[M, Na]^T B[M, Nb], colsum(mask * A)
#include <stdlib.h>
#include <type_traits>
#include "proj_common.h"

namespace {
using namespace proj;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kStepB = 2 * kFrag;   // one 32-column tile of one 32-deep K step: two 16-deep MFMA fragments

__device__ __forceinline__ float bf_bits_to_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ unsigned short bf16_bits(const __bf16 *p) { return *reinterpret_cast<const unsigned short *>(p); }

// ---------------------------------------------------------------------------------------------------
// weight image: for the 32-deep K step s, 32-column tile n32 of the OUTPUT and 16-deep half p:
//   fragment ((s * Np/32 + n32) * 2 + p), 1 KiB, lane (r = l & 31, h = l >> 5) holds
//   B[n32 * 32 + r][32 s + 16 p + 8 h + 0..7]       with B[n][k] = W[n * stride_n + k * stride_k]
// = the B operand of v_mfma_f32_32x32x16_bf16 in lane order; zero-padded to (N to 128, K to 64).
struct ImageJobB {
  const __bf16 *W;
  int64_t sn, sk;
  int N, K;
  char *img;
};
struct ImageJobsB {
  ImageJobB j[8];
};
__global__ void weight_image_bf16_kernel(ImageJobsB jobs) {
  const ImageJobB jb = jobs.j[blockIdx.y];
  const int Np = (jb.N + 127) / 128 * 128, Kp = (jb.K + 63) / 64 * 64;
  const int k8s = Kp / 8;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Np * k8s) return;
  const int n = idx / k8s, k8 = idx - n * k8s;
  unsigned x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    x[j] = (n < jb.N && 8 * k8 + j < jb.K) ? bf16_bits(jb.W + (int64_t)n * jb.sn + (int64_t)(8 * k8 + j) * jb.sk) : 0u;
  const u32x4 v = {x[0] | (x[1] << 16), x[2] | (x[3] << 16), x[4] | (x[5] << 16), x[6] | (x[7] << 16)};
  const int s = k8 >> 2, p = (k8 >> 1) & 1, h = k8 & 1, n32 = n >> 5, r = n & 31;
  char *dst = jb.img + ((size_t)(s * (Np / 32) + n32) * 2 + p) * kFrag + (32 * h + r) * 16;
  *reinterpret_cast<u32x4 *>(dst) = v;
}

// ---------------------------------------------------------------------------------------------------
// proj_rows_bf16: 128 x BN output tiles per 256-thread workgroup (2 x 2 waves, each 64 x BN/2), two workgroups per
// CU, PERSISTENT over a strided tile list whose consecutive slots on one XCD are the column tiles of one row tile
// (its rows come from that XCD's L2 after the first read).  K runs in steps of 32 (two MFMAs per accumulator):
//   W  a RING of three step buffers (BN/32 x 2 KiB each): the image's fragments of one step, a linear copy,
//      requested TWO steps ahead (first version: one step ahead -- 2 400 cycles per step for 2 x 512 cycles of MFMA,
//      every step waited for the L2 round trip of its own fragments)
//   A  block buffers Ab[2] (128 rows x 128 bytes): a 64-deep line block of the tile's rows, piece q = rows
//      8q .. 8q+7 as whole 128-byte lines (8 lines per DMA instruction); LDS slot (row, c) holds the line's 16-byte
//      chunk c ^ ((row >> 1) & 7), so the fragment read of lane (r, h) -- chunk 2 pp + h of row r -- is conflict-free
//      for ds_read_b128's lane groups
// = 80 KiB per workgroup at BN = 256: two of them are exactly the CU's LDS (the row flags of the epilogue live in the
// W buffer the tile's last step has freed).  One barrier per step.  Issue order / waits of a wave (kWP, kAP = its DMA
// pieces per W step / A block):
//   even step e: wait vmcnt(kWP)       [W(e), A(block) landed; W(e+1) stays in flight], barrier,
//                issue W(e+2) x kWP then A(block+1) x kAP, multiply step e
//   odd  step o: wait vmcnt(kWP + kAP) [W(o) landed; W(o+1), A(block+1) stay in flight], barrier,
//                issue W(o+2) x kWP, multiply step o
// The last two steps of a tile request the first two W steps / the first A block of the workgroup's NEXT tile; the
// store tail waits for them (vmcnt(0)) BEFORE its first store, so the first block of the next tile opens with bare
// barriers and the stores retire behind its multiplies (the first counted wait after them is two steps later).
struct RowsArgsB {
  const __bf16 *A;
  int64_t lda;
  int64_t M;
  int K, N;
  const char *wimg;
  const __bf16 *bias;         // [N] or null
  const int32_t *rowptr;      // null: no mask; else rows of nodes with an empty CSR segment come out 0
  int L;
  __bf16 *out;
  int64_t ldc;
  int row_tiles;              // ceil(M / 128); with a node list: ceil(n_nodes / npt)
  int64_t tiles;              // tile slots: row tiles rounded up to 8, times column tiles
  int Kp, Np;                 // K, N padded to multiples of 64 / 128 (= the weight image's shape)
  // GATHER: only the rows of the listed nodes are multiplied and written (R-MAT graphs: 38-48 % of cfg5's nodes have no
  // edge at all on the side a product serves; the reference, which works per EDGE, never touches them either).  A tile
  // covers npt = 128 / L WHOLE nodes (L >= 16: at most 8), row i of tile t = token i % L of node nodes[t npt + i / L]
  const int32_t *nodes;
  int64_t n_nodes;
  int npt;
};

// RAGGED: K % 64 != 0 or N % BN != 0: line chunks beyond K are fetched from the row's first chunk (finite data times
// the image's zero padding), columns beyond N are computed on zero weights and not stored
template <int BN, bool RAGGED, bool GATHER>
__global__ __launch_bounds__(256, 2) void proj_rows_bf16_kernel(RowsArgsB a) {
  constexpr int BM = 128, NW = 4, WN = 2, MTW = 2, NTW = BN / 64;
  constexpr int kWB = BN / 32 * kStepB;        // one W step buffer
  constexpr int kAB = BM * 128;                // one A block buffer
  constexpr int kWP = kWB / kFrag / NW;        // W pieces per wave and step
  constexpr int kAP = kAB / kFrag / NW;        // A pieces per wave and block
  constexpr int NWB = 3;                       // W ring
  constexpr int oA = NWB * kWB;
  static_assert(kWP >= 1 && kAP == 4 && NW * 4096 <= kAB && 2 * BM * 4 <= kWB, "tile / wave shape");
  __shared__ __attribute__((aligned(16))) char smem[oA + 2 * kAB];     // ONE object: [W ring][Ab0][Ab1]

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w / WN, wn = w % WN;
  const int nct = a.Np / BN;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char *)smem;
  const size_t wstep = (size_t)(a.Np / 32) * kStepB;            // bytes of one K step of the whole image

  // A DMA: piece w + NW jj = rows 8 w + (lane >> 3) + 32 jj of the tile, LDS slot lane & 7
  const int rbase = 8 * w + (lane >> 3);
  const int cg8 = 8 * ((lane & 7) ^ ((rbase >> 1) & 7));        // element offset of the chunk this lane fetches
  const __bf16 *arow[kAP];
  // fragment reads: lane (fr, fh); A chunk 2 pp + fh of row fr, swizzled
  const int fr = lane & 31, fh = lane >> 5;
  int ard[4];
#pragma unroll
  for (int pp = 0; pp < 4; ++pp) ard[pp] = oA + (32 * MTW * wm + fr) * 128 + 16 * ((2 * pp + fh) ^ ((fr >> 1) & 7));
  const int brd = (NTW * wn) * kStepB + lane * 16;

  struct Tile {
    int64_t row0, rt;
    int col0;
    bool valid;
  };
  auto tile_of = [&](int64_t u) {
    Tile tl;
    const int64_t i_x = u / kXcd;
    const int64_t rt = (i_x / nct) * kXcd + u % kXcd;
    tl.rt = rt;
    tl.row0 = rt * BM;
    tl.col0 = (int)(i_x % nct) * BN;
    tl.valid = u < a.tiles && rt < a.row_tiles;
    return tl;
  };
  // GATHER: actual row of tile row i (or -1 if the tile has no node there); the tile's node ids are wave-uniform loads
  // (explicit s_load: hipcc reads a list that might alias the output through the VECTOR memory path, and the wait it
  // then inserts for those loads -- vmcnt(0) -- would drain this kernel's DMA queue every tile; the scalar path counts on
  // lgkmcnt.  The list is padded by 8 entries (include/ampconv.h), slots beyond n_nodes are read and not used)
  auto gather_row = [&](const Tile &tl, int i) -> int64_t {
    const int64_t k0 = tl.rt * a.npt;
    int nid[8];
    {
      const int32_t *np = a.nodes + k0;
      asm volatile("s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %8, 0x4\n\ts_load_dword %2, %8, 0x8\n\t"
                   "s_load_dword %3, %8, 0xc\n\ts_load_dword %4, %8, 0x10\n\ts_load_dword %5, %8, 0x14\n\t"
                   "s_load_dword %6, %8, 0x18\n\ts_load_dword %7, %8, 0x1c\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(nid[0]), "=&s"(nid[1]), "=&s"(nid[2]), "=&s"(nid[3]), "=&s"(nid[4]), "=&s"(nid[5]), "=&s"(nid[6]),
                     "=&s"(nid[7])
                   : "s"(np)
                   : "memory");
    }
    const int k = i / a.L, l = i - k * a.L;
    int node = nid[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) node = k == q ? nid[q] : node;
    return (k < a.npt && k0 + k < a.n_nodes) ? (int64_t)node * a.L + l : -1;
  };
  auto rebase = [&](const Tile &tl) {
#pragma unroll
    for (int jj = 0; jj < kAP; ++jj) {
      int64_t m;
      if (GATHER) {
        m = gather_row(tl, rbase + 32 * jj);
        m = m >= 0 ? m : 0;                                                     // any valid row (its product is not stored)
      } else {
        m = tl.row0 + rbase + 32 * jj;
        m = m < a.M ? m : a.M - 1;
      }
      arow[jj] = a.A + m * a.lda;
    }
  };
  // ONE DMA piece of an A block / a W step (the main loop hands them out between its MFMAs; the prologue takes whole sets)
  auto issue_a_piece = [&](int kb, int abuf, int jj) {
    int off = kb * 64 + cg8;
    if (RAGGED) off = off < a.K ? off : 0;
    dma16(arow[jj] + off, lds0 + oA + abuf * kAB + (w + NW * jj) * kFrag);
  };
  auto issue_w_piece = [&](int col0, int step, int wbuf, int j) {
    const char *src = a.wimg + (size_t)(col0 / 32) * kStepB + (size_t)step * wstep + (w + NW * j) * kFrag + lane * 16;
    dma16(src, lds0 + wbuf * kWB + (w + NW * j) * kFrag);
  };
  f32x16 acc[MTW][NTW];
  // One 32-deep step: 2 x MTW x NTW MFMAs.  The step's NPc DMA pieces are issued BETWEEN the MFMAs, one every
  // NM / NPc products (pinned with sched_barrier): issued in one burst behind the barrier -- eight waves at once -- every
  // piece waits for the address path of all the others (~150 cycles each, in front of the step's first fragment read)
  auto multiply = [&](auto par_c, int woff, int aoff, auto npc_c, auto &&piece) {
    constexpr int par = decltype(par_c)::value, NPc = decltype(npc_c)::value;
    constexpr int NM = 2 * MTW * NTW;
    i32x4 af[2][MTW], bf[2][NTW];
    auto fragments = [&](auto p_c) {
      constexpr int p = decltype(p_c)::value;
#pragma unroll
      for (int i = 0; i < MTW; ++i)
        af[p][i] = *reinterpret_cast<const i32x4 *>(smem + aoff + ard[2 * par + p] + i * 4096);
#pragma unroll
      for (int j = 0; j < NTW; ++j)
        bf[p][j] = *reinterpret_cast<const i32x4 *>(smem + woff + brd + j * kStepB + p * kFrag);
    };
    fragments(std::integral_constant<int, 0>{});
    int c = 0;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
#ifdef AMPCONV_PROJ_ABLATE_MFMA        // timing-only developer build: fragments read, nothing multiplied
          asm volatile("" ::"v"(af[p][i]), "v"(bf[p][j]));
#else
          acc[i][j] = MFMA32(af[p][i], bf[p][j], acc[i][j]);
#endif
          ++c;
          if (c == NM / 4) fragments(std::integral_constant<int, 1>{});      // second half's fragments, a quarter ahead
#pragma unroll
          for (int k = 0; k < NPc; ++k)
            if (c == k * NM / NPc + 1) {
              __builtin_amdgcn_sched_barrier(0);
              piece(k);
              __builtin_amdgcn_sched_barrier(0);
            }
        }
  };

  int64_t u = blockIdx.x;
  Tile cur = tile_of(u);
  if (!cur.valid) return;        // slots are ordered and the grid is a multiple of 8: nothing further either
  Tile nxt = tile_of(u + gridDim.x);
  const int KB = a.Kp / 64;
  int ab = 0;                    // A buffer of the block about to be multiplied
  int wb = 0;                    // W ring slot of the step about to be multiplied
  rebase(cur);
#pragma unroll
  for (int j = 0; j < kWP; ++j) issue_w_piece(cur.col0, 0, 0, j);
#pragma unroll
  for (int j = 0; j < kWP; ++j) issue_w_piece(cur.col0, 1, 1, j);
#pragma unroll
  for (int jj = 0; jj < kAP; ++jj) issue_a_piece(0, 0, jj);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // W step `st` (counted from the current tile's first step) -> its tile's first column and its step there
  auto w_col = [&](int st) { return st < 2 * KB ? cur.col0 : (nxt.valid ? nxt.col0 : cur.col0); };
  auto w_step = [&](int st) { return st < 2 * KB ? st : st - 2 * KB; };
  auto ring = [](int x) { return x >= NWB ? x - NWB : x; };

  for (;;) {
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int rp0 = 0, rp1 = 1;          // CSR bounds of this thread's row (mask) and raw bias entries of its columns:
    unsigned braw[NTW];            // requested in the tile's last step, first USED behind the tail's wait
    for (int kb = 0; kb < KB; ++kb) {
      const bool last = kb + 1 == KB;
      // ---- even step: A halves 0, 1 of Ab[ab].  (kb == 0: the tail of the previous tile, or the prologue, has
      // already waited for both steps of this block)
      if (kb != 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWP) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (last) rebase(nxt.valid ? nxt : cur);      // past the end of the work list the refill is harmless
      {
        const int wc = w_col(2 * kb + 2), ws = w_step(2 * kb + 2), wr = ring(wb + 2), an = last ? 0 : kb + 1;
        multiply(std::integral_constant<int, 0>{}, wb * kWB, ab * kAB, std::integral_constant<int, kWP + kAP>{},
                 [&](int k) {
                   if (k < kWP) issue_w_piece(wc, ws, wr, k);
                   else issue_a_piece(an, ab ^ 1, k - kWP);
                 });
      }
      wb = ring(wb + 1);
      // ---- odd step: A halves 2, 3
      if (kb != 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWP + kAP) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const int wc1 = w_col(2 * kb + 3), ws1 = w_step(2 * kb + 3), wr1 = ring(wb + 2);
      if (last) {
        // epilogue operands of THIS tile: ordinary loads issued after every DMA of the step; their values pass
        // through an empty asm in the tail, so the compiler's wait for them sits there and not here
        // (no control flow around them either: a load inside a branch gets its wait at the join.  Without a mask /
        // bias the loads read the weight image and their values are not used.  M < 2^31 with a mask: host check)
        {
          const int64_t m = cur.row0 + (t & (BM - 1)) < a.M ? cur.row0 + (t & (BM - 1)) : a.M - 1;
          const unsigned node = (unsigned)m / (unsigned)a.L;
          const int32_t *rpp = a.rowptr ? a.rowptr + node : reinterpret_cast<const int32_t *>(a.wimg);
          rp0 = rpp[0];
          rp1 = rpp[1];
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          int col = cur.col0 + (NTW * wn + j) * 32 + fr;
          if (RAGGED) col = col < a.N ? col : a.N - 1;
          const __bf16 *bp = a.bias ? a.bias + col : reinterpret_cast<const __bf16 *>(a.wimg);
          braw[j] = (unsigned)bf16_bits(bp);
        }
      }
      multiply(std::integral_constant<int, 1>{}, wb * kWB, ab * kAB, std::integral_constant<int, kWP>{},
               [&](int k) { issue_w_piece(wc1, ws1, wr1, k); });
      wb = ring(wb + 1);
      ab ^= 1;
    }
    asm volatile("" : "+v"(rp0), "+v"(rp1));
    // ---- store tail.  Everything this wave has requested (first block of the next tile) must have landed before the
    // stores go out: the next block then needs no wait of its own.  The row flags go through the W slot the last step
    // was multiplied from (ring slot wb + 2: free once every wave has passed the barrier).  C/D register e of lane
    // (col = lane & 31, hi = lane >> 5) is row (e & 3) + 8 (e >> 2) + 4 hi of a 32 x 32 tile: bias and row mask applied
    // in that layout, then the tile goes through a wave-private 4 KiB of the A buffer the last block was multiplied
    // from (the next tile's block 0 went to the other one) as fp32 and leaves as bf16, 16 rows x 64 bytes per store
    // instruction.
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float *flags = reinterpret_cast<float *>(smem + ring(wb + 2) * kWB);
    int *rowtab = reinterpret_cast<int *>(flags + BM);        // GATHER: actual output row of every tile row, or -1
    if (a.rowptr) flags[t & (BM - 1)] = rp1 != rp0 ? 1.f : 0.f;      // both halves of the workgroup write the same values
    if (GATHER) rowtab[t & (BM - 1)] = (int)gather_row(cur, t & (BM - 1));
    if (a.rowptr || GATHER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    {
      float *stage = reinterpret_cast<float *>(smem + oA + (ab ^ 1) * kAB + w * 4096);
      const int sr = lane >> 2, sc = lane & 3;
      float bj[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        asm volatile("" : "+v"(braw[j]));
        bj[j] = a.bias ? bf_bits_to_f32((unsigned short)braw[j]) : 0.f;   // (padded columns are not stored)
      }
      auto store_tile = [&](auto ragged_rows) {
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
          const int rl0 = (MTW * wm + i) * 32;
          float fl[16];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float4 f4 = make_float4(1.f, 1.f, 1.f, 1.f);
            if (a.rowptr) f4 = *reinterpret_cast<const float4 *>(flags + rl0 + 4 * fh + 8 * g);
            fl[4 * g] = f4.x; fl[4 * g + 1] = f4.y; fl[4 * g + 2] = f4.z; fl[4 * g + 3] = f4.w;
          }
#pragma unroll
          for (int j = 0; j < NTW; ++j) {
            const int colt = cur.col0 + (NTW * wn + j) * 32;
#pragma unroll
            for (int e = 0; e < 16; ++e)
              stage[((e & 3) + 8 * (e >> 2) + 4 * fh) * 32 + fr] = (acc[i][j][e] + bj[j]) * fl[e];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const int row = sr + 16 * g;
              const float4 v0 = *reinterpret_cast<const float4 *>(stage + row * 32 + 8 * sc);
              const float4 v1 = *reinterpret_cast<const float4 *>(stage + row * 32 + 8 * sc + 4);
              const u32x4 pk = {cvt_pk_bf16(v0.x, v0.y), cvt_pk_bf16(v0.z, v0.w), cvt_pk_bf16(v1.x, v1.y),
                                cvt_pk_bf16(v1.z, v1.w)};
#ifdef AMPCONV_PROJ_ABLATE_STORES      // timing-only developer build: everything but the global stores
              asm volatile("" ::"v"(pk));
#else
              if (GATHER) {
                const int orow = rowtab[rl0 + row];
                if (orow >= 0 && (!RAGGED || colt + 8 * sc < a.N))
                  *reinterpret_cast<u32x4 *>(a.out + (int64_t)orow * a.ldc + colt + 8 * sc) = pk;
              } else if ((!decltype(ragged_rows)::value || cur.row0 + rl0 + row < a.M) && (!RAGGED || colt + 8 * sc < a.N))
                *reinterpret_cast<u32x4 *>(a.out + (cur.row0 + rl0 + row) * a.ldc + colt + 8 * sc) = pk;
#endif
            }
          }
        }
      };
      if (GATHER || cur.row0 + BM <= a.M)  // workgroup-uniform: only the last row tile is ragged (GATHER: the row table masks)
        store_tile(std::false_type{});
      else
        store_tile(std::true_type{});
    }
    if (!nxt.valid) break;
    cur = nxt;
    u += gridDim.x;
    nxt = tile_of(u + gridDim.x);
  }
}

// ---------------------------------------------------------------------------------------------------
// proj_wgrad_bf16: dW[Na, Nb] = sum over the rows m of A[m, :]^T B[m, :] (+ column sums of mask * A).
// The contraction index is the ROW of both inputs, so an MFMA operand fragment (8 consecutive k per lane) is a
// COLUMN piece of the row-major tiles: the row tiles are copied as they are, [32 rows][T columns] bf16 with the
// 64-byte block index XORed by (row & 3) (on the DMA's source address), and read with ds_read_b64_tr_b16 (the layout
// and the read addresses of the fp32 kernel, proj_gemm.hip).  T x T tile of dW per workgroup and row slice (T = 256:
// 8 waves, one workgroup per CU; T = 128: 4 waves, two per CU); 32 rows per stage, FOUR stage buffers: three stages
// (96 KiB per CU) are in flight while one is multiplied -- the product is HBM-bound (1 KiB of rows per 0.13 MFLOP at
// T = 256), so what matters is bytes in flight, not registers.  One barrier per stage:
//   stage s: wait vmcnt(2 x (pieces + 1)) [stage s landed; s+1, s+2 stay in flight], barrier, issue s+3, multiply s
// Per-row flags (1.0 / 0.0: row inside the slice AND node has an in-edge; row_flags_kernel) travel with the stage
// (64 bytes); they act on the column sums (VALU, from the LDS image).  Rows beyond the slice and columns beyond Na /
// Nb are fetched from a zero row.  Partial tiles per slice go to the workspace and are added in slice order
// (bitwise reproducible, no atomics).
struct WgradArgsB {
  const __bf16 *A;
  int64_t lda;
  const __bf16 *B;
  int64_t ldb;
  int64_t M;
  int Na, Nb;
  float *part;                // [S][Nap * Nbp + Nap]
  int S;
  int64_t rows_per_slice;     // multiple of 32
  int Nap, Nbp;               // Na, Nb padded to multiples of 128
  const __bf16 *flags;        // [S * rows_per_slice]
  const __bf16 *zrow;         // 512 bytes of zeros
  int want_colsum;
  // GATHER: the sum runs over the rows of the listed nodes only.  rowmap[mc] = actual row of compact row mc (token mc % L
  // of node nodes[mc / L]; -1 beyond the list), written with the flags by row_flags_kernel; the 32 entries of a stage
  // travel with the DMA group issued three stages before that stage's own rows are requested
  const int32_t *rowmap;      // [S * rows_per_slice + 6 * 32] or null
};

constexpr int kRSB = 32;      // rows per stage

template <int T, bool GATHER>
__global__ __launch_bounds__(T == 256 ? 512 : 256, 2) void proj_wgrad_bf16_kernel(WgradArgsB a) {
  constexpr int NW = T == 256 ? 8 : 4, NTHR = 64 * NW, WI = 2, WJ = NW / WI;
  constexpr int kRow = T * 2;                       // bytes per image row
  constexpr int kImg = kRSB * kRow;                 // one operand's stage image
  constexpr int kStage = 2 * kImg;
  constexpr int NBUF = 4;
  constexpr int NIW = T / 32 / WI, NJW = T / 32 / WJ;
  constexpr int kRPP = kFrag / kRow;                // rows per DMA piece: 2 / 4
  constexpr int kPieces = kImg / kFrag;             // pieces per operand and stage: 16 / 8
  constexpr int kPW = 2 * kPieces / NW;             // pieces per wave and stage: 4
  constexpr int CPR = kRow / 16;                    // 16-byte chunks per row: 32 / 16
  constexpr int oFl = NBUF * kStage;                // per stage slot: 64 B of row flags, 128 B of row map
  constexpr int kMeta = 192;
  static_assert(kPW == 4 && kPieces % NW == 0 && NTHR / (T / 8) == 16, "tile / wave shape");
  __shared__ __attribute__((aligned(16))) char smem[oFl + NBUF * kMeta];

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wi = w / WJ, wj = w % WJ;
  const int ntj = a.Nbp / T, ntiles = (a.Nap / T) * ntj;
  const int b = blockIdx.x, xcd = b % kXcd, i_x = b / kXcd;
  const int slice = (i_x / ntiles) * kXcd + xcd, tile = i_x % ntiles;
  if (slice >= a.S) return;
  const int ti = tile / ntj, tj = tile % ntj;
  const int64_t m0 = (int64_t)slice * a.rows_per_slice;
  const int64_t m1 = m0 + a.rows_per_slice < a.M ? m0 + a.rows_per_slice : a.M;
  const int ns = (int)((m1 - m0 + kRSB - 1) / kRSB);
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char *)smem;

  // DMA pieces of this wave: w + NW jj; jj < kPW / 2: operand A, else B.  Lane = (row lrow of the piece, slot cs)
  const int lrow = lane / CPR, cs = lane % CPR;
  const __bf16 *src[kPW];      // this lane's source in stage 0 (advanced 32 rows per stage); zero row if its column is padding
  int64_t mrow[kPW];           // its row in stage 0
  bool colok[kPW];
  const __bf16 *zsrc = a.zrow + 8 * (lane & 31);
#pragma unroll
  for (int jj = 0; jj < kPW; ++jj) {
    const bool isb = jj >= kPW / 2;
    const int pc = w + NW * (jj % (kPW / 2));                  // piece index within the operand
    const int r = pc * kRPP + lrow;                             // row within the stage
    const int c = cs ^ ((r & 3) << 2);                          // global chunk that belongs at LDS slot cs
    const int col = (isb ? tj : ti) * T + 8 * c;
    colok[jj] = col < (isb ? a.Nb : a.Na);
    mrow[jj] = m0 + r;
    src[jj] = (isb ? a.B : a.A) + col;
  }
  const int64_t ld[2] = {a.lda, a.ldb};
  // DMA group g = the rows of stage g, its 32 row flags and (GATHER) the row map of stage g + 3 -- the stage whose rows
  // are requested three groups later.  Flags and map are ONE instruction (lanes 0..3 / 4..11) into the meta slot g & 3:
  // [64 B flags of stage g][128 B map of stage g + 3]; so the map of group g sits in slot (g - 3) & 3 = (g + 1) & 3.  Every
  // wave copies the same bytes to the same place: every wave issues the same number of vector-memory instructions per
  // group (4 pieces + 1; the waits below are counted).
  auto issue = [&](int s) {
    const int buf = s & (NBUF - 1);
    const unsigned dst = lds0 + buf * kStage + w * kFrag;
    const int *map = reinterpret_cast<const int *>(smem + oFl + ((s + 1) & (NBUF - 1)) * kMeta + 64);
#pragma unroll
    for (int jj = 0; jj < kPW; ++jj) {
      const bool isb = jj >= kPW / 2;
      int64_t m = mrow[jj] + (int64_t)s * kRSB;
      bool rowok = m < m1;
      if (GATHER) {
        const int am = map[(int)(mrow[jj] - m0)];       // mrow - m0 = row within the stage
        rowok = rowok && am >= 0;
        m = am >= 0 ? am : 0;
      }
      const __bf16 *p = (colok[jj] && rowok) ? src[jj] + m * ld[isb ? 1 : 0] : zsrc;
      dma16(p, dst + (isb ? kImg : 0) + NW * (jj % (kPW / 2)) * kFrag);
    }
    if (lane < (GATHER ? 12 : 4)) {
      const char *sp = (!GATHER || lane < 4)
                           ? reinterpret_cast<const char *>(a.flags + m0 + (int64_t)s * kRSB) + 16 * lane
                           : reinterpret_cast<const char *>(a.rowmap + m0 + (int64_t)(s + 3) * kRSB) + 16 * (lane - 4);
      dma16(sp, lds0 + oFl + buf * kMeta);
    }
  };
  if (GATHER) {
    // bootstrap: the maps of stages 0, 1, 2 (the three groups of the prologue), into the slots the pipeline would use
#pragma unroll
    for (int g0 = 0; g0 < 3; ++g0)
      if (lane >= 4 && lane < 12)
        dma16(reinterpret_cast<const char *>(a.rowmap + m0 + (int64_t)g0 * kRSB) + 16 * (lane - 4),
              lds0 + oFl + ((g0 + 1) & (NBUF - 1)) * kMeta);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // transposed reads: lane = (h = lane >> 5, gi = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3) addresses
  // row 8 h + 4 u + q, columns 32 tile + 16 gi + 4 p .. + 3 (u = 0, 1: the two halves of the 8-deep k group)
  const int fh = lane >> 5, gi = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3, fr = lane & 31;
  const int rd = (8 * fh + q) * kRow + (q << 6) + 32 * gi + 8 * pp;      // ^ (tile & 3) << 6, + (tile >> 2) << 8

  f32x16 acc[NIW][NJW];
#pragma unroll
  for (int i = 0; i < NIW; ++i)
#pragma unroll
    for (int j = 0; j < NJW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float cs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool do_cs = a.want_colsum && tj == 0;           // workgroup-uniform
  const int ccol = t % (T / 8), crg = t / (T / 8);       // column sums: 8 columns, rows 2 crg, 2 crg + 1 of a stage

  issue(0);
  issue(1);
  issue(2);
  for (int s = 0; s < ns; ++s) {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(2 * (kPW + 1)) : "memory");
    issue(s + 3);        // into the buffer stage s - 1 was multiplied from (past the end: zero rows, harmless)
    const char *buf = smem + (s & (NBUF - 1)) * kStage;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int so = sub * 16 * kRow;
      i32x4 bf[NJW], af[NIW];
#pragma unroll
      for (int j = 0; j < NJW; ++j) {
        const int jt = NJW * wj + j;
        const char *r0 = buf + kImg + so + ((rd + ((jt >> 2) << 8)) ^ ((jt & 3) << 6));
        bf[j] = tr_frag(r0, r0 + 4 * kRow);
      }
#pragma unroll
      for (int i = 0; i < NIW; ++i) {
        const int it = NIW * wi + i;
        const char *r0 = buf + so + ((rd + ((it >> 2) << 8)) ^ ((it & 3) << 6));
        af[i] = tr_frag(r0, r0 + 4 * kRow);
      }
      // every transposed fragment of the half stage is in its registers before the first MFMA issues, and none is read in
      // among the MFMAs (csrc/proj_gemm.hip, proj_wgrad_kernel: the schedule hipcc picks by itself gave wrong sums there)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NIW; ++i)
#pragma unroll
        for (int j = 0; j < NJW; ++j) acc[i][j] = MFMA32(af[i], bf[j], acc[i][j]);
      // (... and the next half stage's reads do not redefine a fragment register right behind the MFMA that reads it:
      // tools/scan_tr_hazard.py --gate, rule WAR)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 7" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    if (do_cs) {
      const char *fl = smem + oFl + (s & (NBUF - 1)) * kMeta;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * crg + rr;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(buf + r * kRow + ((16 * ccol) ^ ((r & 3) << 6)));
        const bool on = *reinterpret_cast<const unsigned short *>(fl + 2 * r) != 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          cs8[2 * k] += on ? lo_as_f32(v[k]) : 0.f;
          cs8[2 * k + 1] += on ? hi_as_f32(v[k]) : 0.f;
        }
      }
    }
  }
  // nothing of this wave may still be on its way into LDS when the buffers are reused below or the workgroup ends
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // partial tile of this slice
  float *part = a.part + (size_t)slice * ((size_t)a.Nap * a.Nbp + a.Nap);
#pragma unroll
  for (int i = 0; i < NIW; ++i)
#pragma unroll
    for (int j = 0; j < NJW; ++j) {
      float *o = part + (size_t)(ti * T + (NIW * wi + i) * 32 + 4 * fh) * a.Nbp + tj * T + (NJW * wj + j) * 32 + fr;
#pragma unroll
      for (int e = 0; e < 16; ++e) o[(size_t)((e & 3) + 8 * (e >> 2)) * a.Nbp] = acc[i][j][e];
    }
  if (tj == 0) {
    // column sums of the A tile: 16 row groups per 8-column chunk, added through LDS in a fixed order
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float4 *red = reinterpret_cast<float4 *>(smem);
    red[(crg * (T / 8) + ccol) * 2] = make_float4(cs8[0], cs8[1], cs8[2], cs8[3]);
    red[(crg * (T / 8) + ccol) * 2 + 1] = make_float4(cs8[4], cs8[5], cs8[6], cs8[7]);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t < T / 4) {
      float4 sum = red[t];
#pragma unroll
      for (int r = 1; r < 16; ++r) {
        const float4 v = red[r * (T / 4) + t];
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
      }
      *reinterpret_cast<float4 *>(part + (size_t)a.Nap * a.Nbp + ti * T + 4 * t) = sum;
    }
  }
}

// flags[m] = 1.0 if row m < M and (no mask or the row's node has an in-edge) else 0.0, for m < Mpad (a multiple of 8);
// the first workgroup also clears the zero row
__global__ __launch_bounds__(256) void row_flags_kernel(const int32_t *__restrict__ rowptr, int L, int64_t M, int64_t Mpad,
                                                        unsigned short *__restrict__ flags, u32x4 *__restrict__ zrow,
                                                        const int32_t *__restrict__ nodes, int32_t *__restrict__ rowmap) {
  if (blockIdx.x == 0 && threadIdx.x < 32) zrow[threadIdx.x] = u32x4{0u, 0u, 0u, 0u};
  const int64_t m8 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (rowmap && m8 < Mpad + 6 * 32) {          // the pipeline reads the map up to six stages past a slice's end
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t m = m8 + k;
      rowmap[m] = m < M ? (int32_t)((int64_t)nodes[m / L] * L + m % L) : -1;
    }
  }
  if (m8 >= Mpad) return;
  unsigned f[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t m = m8 + k;
    bool on = m < M;
    if (on && rowptr) {
      const int64_t node = nodes ? nodes[m / L] : m / L;
      on = rowptr[node + 1] != rowptr[node];
    }
    f[k] = on ? 0x3F80u : 0u;
  }
  *reinterpret_cast<u32x4 *>(flags + m8) = u32x4{f[0] | (f[1] << 16), f[2] | (f[3] << 16), f[4] | (f[5] << 16),
                                                 f[6] | (f[7] << 16)};
}

// out[e] = bf16(sum over the slices of part[s][e]); the fp32 kernel's reduction (proj_gemm.hip) with a bf16 epilogue
__global__ __launch_bounds__(256) void wgrad_reduce_bf16_kernel(const float *__restrict__ part, int S, int Na, int Nb,
                                                                int Nap, int Nbp, unsigned short *__restrict__ dW,
                                                                unsigned short *__restrict__ colsum) {
  __shared__ float4 red[8][32];
  const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int64_t n_dw = (int64_t)Nap * Nbp, n_all = n_dw + Nap;
  const int64_t e = ((int64_t)blockIdx.x * 32 + el) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < n_all) {
    float4 a1 = acc;
    int s = g;
    for (; s + 8 < S; s += 16) {
      const float4 v0 = *reinterpret_cast<const float4 *>(part + (size_t)s * n_all + e);
      const float4 v1 = *reinterpret_cast<const float4 *>(part + (size_t)(s + 8) * n_all + e);
      acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
    }
    if (s < S) {
      const float4 v0 = *reinterpret_cast<const float4 *>(part + (size_t)s * n_all + e);
      acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
    }
    acc.x += a1.x; acc.y += a1.y; acc.z += a1.z; acc.w += a1.w;
  }
  red[g][el] = acc;
  __syncthreads();
  if (g == 0 && e < n_all) {
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const float4 v = red[k][el];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const i32x2 pk = {(int)cvt_pk_bf16(acc.x, acc.y), (int)cvt_pk_bf16(acc.z, acc.w)};
    if (e < n_dw) {
      const int i = (int)(e / Nbp), j = (int)(e - (int64_t)i * Nbp);
      if (i < Na && j < Nb) *reinterpret_cast<i32x2 *>(dW + (size_t)i * Nb + j) = pk;
    } else if (colsum && e - n_dw < Na) {
      *reinterpret_cast<i32x2 *>(colsum + (e - n_dw)) = pk;
    }
  }
}

struct WgradPlanB {
  int S;
  int64_t rows_per_slice;
  int T;
};
inline WgradPlanB wgrad_plan_bf16(int64_t M, int Nap, int Nbp) {       // padded shape (multiples of 128)
  WgradPlanB p;
  static const bool small_t = [] {                        // developer switch: 128 x 128 tiles of four waves, two per CU
    const char *e = getenv("AMPCONV_PROJ_WGRAD_BF16_T");
    return e && atoi(e) == 128;
  }();
  p.T = (Nap % 256 == 0 && Nbp % 256 == 0 && !small_t) ? 256 : 128;
  const int64_t ntiles = (int64_t)(Nap / p.T) * (Nbp / p.T);
  const int64_t nstages = (M + kRSB - 1) / kRSB;
  // one round of workgroups: slices are dealt to the 8 XCDs in turn and every slice brings `ntiles` workgroups, so
  // an XCD's 32 CUs (x 2 for the 4-wave shape) hold floor(32 / ntiles) slices each (proj_gemm.hip: wgrad_plan)
  const int per_xcd = (p.T == 256 ? 1 : 2) * 32;
  int64_t S = (int64_t)kXcd * (per_xcd / ntiles > 0 ? per_xcd / ntiles : 1);
  if (S > nstages) S = nstages > 0 ? nstages : 1;
  p.rows_per_slice = ((nstages + S - 1) / S) * kRSB;
  p.S = (int)((M + p.rows_per_slice - 1) / p.rows_per_slice);
  if (p.S < 1) p.S = 1;
  return p;
}

inline size_t align16(size_t x) { return (x + 15) / 16 * 16; }

}  // namespace

size_t ampconv_proj_weight_image_bytes_bf16(int N, int K) {
  if (N <= 0 || K <= 0) return 0;
  return (size_t)((N + 127) / 128 * 128) * (size_t)((K + 63) / 64 * 64) * 2;
}

bool ampconv_proj_supported_bf16(int N, int K) { return N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0; }

int ampconv_proj_weight_images_bf16(int count, const ampconv_weight_image_t *jobs, hipStream_t stream) {
  ImageJobsB js;
  int most = 0;
  for (int i = 0; i < count; ++i) {
    const ampconv_weight_image_t &w = jobs[i];
    if (!ampconv_proj_supported_bf16(w.N, w.K) || !w.W || !w.image || (uintptr_t)w.image % 16) return AMPCONV_E_BADARG;
    js.j[i] = ImageJobB{(const __bf16 *)w.W, w.stride_n, w.stride_k, w.N, w.K, (char *)w.image};
    const int total = ((w.N + 127) / 128 * 128) * (((w.K + 63) / 64 * 64) / 8);
    most = total > most ? total : most;
  }
  weight_image_bf16_kernel<<<dim3((most + 255) / 256, count), 256, 0, stream>>>(js);
  return ampconv_launch_status();
}

int ampconv_proj_rows_bf16(const void *A, int64_t lda, int64_t M, int K, const void *wimage, int N, const void *bias,
                           const int32_t *rowptr, int L, void *out, int64_t ldc, const int32_t *nodes, int64_t n_nodes,
                           hipStream_t stream) {
  if (M < 0 || !ampconv_proj_supported_bf16(N, K) || lda < K || ldc < N || lda % 8 || ldc % 8) return AMPCONV_E_BADARG;
  if (M == 0) return AMPCONV_OK;
  if (!A || !wimage || !out || (uintptr_t)A % 16 || (uintptr_t)wimage % 16 || (uintptr_t)out % 16)
    return AMPCONV_E_BADARG;
  if (rowptr && (L <= 0 || M > 0x7fffffff)) return AMPCONV_E_BADARG;
  // node list: whole nodes per tile (L in [16, 128]), no mask (list the nodes that pass it), rows addressed as int32
  if (nodes && (rowptr || L < 16 || L > 128 || n_nodes < 0 || n_nodes * L > M || M > 0x7fffffff)) return AMPCONV_E_BADARG;
  if (nodes && n_nodes == 0) return AMPCONV_OK;
  const int Np = (N + 127) / 128 * 128, Kp = (K + 63) / 64 * 64;
  const bool ragged = Np != N || Kp != K;
  const int bn = Np % 256 ? 128 : 256;
  const int npt = nodes ? 128 / L : 0;
  const int64_t rts = nodes ? (n_nodes + npt - 1) / npt : (M + 127) / 128;
  if (rts > (int64_t)INT32_MAX / 64) return AMPCONV_E_BADARG;
  const int64_t rtp = (rts + 7) / 8 * 8;
  RowsArgsB a{(const __bf16 *)A, lda, M, K, N, (const char *)wimage, (const __bf16 *)bias, rowptr, L, (__bf16 *)out,
              ldc, (int)rts, rtp * (Np / bn), Kp, Np, nodes, n_nodes, npt};
  // two workgroups per CU; a multiple of 8 (slot u of a workgroup keeps u % 8, its XCD label)
  int64_t grid = (int64_t)cu_count() * 2 / kXcd * kXcd;
  if (grid < kXcd) grid = kXcd;
  if (grid > a.tiles) grid = a.tiles;
  const unsigned g = (unsigned)grid;
#define ROWS_LAUNCH(BN_, RG_)                                                                    \
  do {                                                                                           \
    if (nodes) proj_rows_bf16_kernel<BN_, RG_, true><<<g, 256, 0, stream>>>(a);                  \
    else proj_rows_bf16_kernel<BN_, RG_, false><<<g, 256, 0, stream>>>(a);                       \
  } while (0)
  if (bn == 256 && !ragged) ROWS_LAUNCH(256, false);
  else if (bn == 256) ROWS_LAUNCH(256, true);
  else if (!ragged) ROWS_LAUNCH(128, false);
  else ROWS_LAUNCH(128, true);
#undef ROWS_LAUNCH
  return ampconv_launch_status();
}

// workspace: [S partial slabs, fp32][zero row, 512 B][row flags, S * rows_per_slice bf16][256 B: the flag requests of
// the three stages a workgroup runs ahead of its last one]
size_t ampconv_proj_wgrad_workspace_bytes_bf16(int64_t M, int Na, int Nb) {
  if (M < 0 || Na <= 0 || Nb <= 0 || Na % 8 || Nb % 8) return 0;
  const int Nap = (Na + 127) / 128 * 128, Nbp = (Nb + 127) / 128 * 128;
  const WgradPlanB p = wgrad_plan_bf16(M, Nap, Nbp);
  // [partials][zero row 512 B][flags + 256 B][row map: (rows + 6 stages) int32, node lists only: sized always]
  const size_t rows = (size_t)p.S * p.rows_per_slice;
  return (size_t)p.S * ((size_t)Nap * Nbp + Nap) * sizeof(float) + 512 + align16(rows * 2) + 256 + (rows + 6 * 32) * 4;
}

int ampconv_proj_wgrad_bf16(const void *A, int64_t lda, const void *B, int64_t ldb, int64_t M, int Na, int Nb,
                            const int32_t *rowptr, int L, void *dW, void *colsum, void *workspace,
                            size_t workspace_bytes, const int32_t *nodes, int64_t n_nodes, hipStream_t st) {
  if (M < 0 || Na <= 0 || Nb <= 0 || Na % 8 || Nb % 8 || lda < Na || ldb < Nb || lda % 8 || ldb % 8)
    return AMPCONV_E_BADARG;
  if (!dW || (uintptr_t)dW % 16 || (colsum && (uintptr_t)colsum % 16)) return AMPCONV_E_BADARG;
  if (rowptr && L <= 0) return AMPCONV_E_BADARG;
  if (nodes && (L < 16 || n_nodes < 0 || n_nodes * L > M || M > 0x7fffffff)) return AMPCONV_E_BADARG;
  const int64_t Me = nodes ? n_nodes * L : M;       // rows the sum runs over (workspace sized for M >= Me)
  if (Me == 0) {
    hipError_t e = hipMemsetAsync(dW, 0, 2 * (size_t)Na * Nb, st);
    if (e == hipSuccess && colsum) e = hipMemsetAsync(colsum, 0, 2 * (size_t)Na, st);
    return e == hipSuccess ? AMPCONV_OK : (int)e;
  }
  if (!A || !B || (uintptr_t)A % 16 || (uintptr_t)B % 16 || !workspace || (uintptr_t)workspace % 16)
    return AMPCONV_E_BADARG;
  const int Nap = (Na + 127) / 128 * 128, Nbp = (Nb + 127) / 128 * 128;
  const WgradPlanB p = wgrad_plan_bf16(Me, Nap, Nbp);
  const size_t n_all = (size_t)Nap * Nbp + Nap;
  const size_t part_bytes = (size_t)p.S * n_all * sizeof(float);
  const int64_t Mpad = (int64_t)p.S * p.rows_per_slice;
  const size_t flag_bytes = align16((size_t)Mpad * 2) + 256;
  if (workspace_bytes < part_bytes + 512 + flag_bytes + (nodes ? ((size_t)Mpad + 6 * 32) * 4 : 0)) return AMPCONV_E_WORKSPACE;
  char *ws = (char *)workspace;
  unsigned short *flags = (unsigned short *)(ws + part_bytes + 512);
  u32x4 *zrow = (u32x4 *)(ws + part_bytes);
  int32_t *rowmap = nodes ? (int32_t *)(ws + part_bytes + 512 + flag_bytes) : nullptr;
  row_flags_kernel<<<(unsigned)(((Mpad + 6 * 32) / 8 + 255) / 256), 256, 0, st>>>(rowptr, L > 0 ? L : 1, Me, Mpad, flags, zrow,
                                                                                 nodes, rowmap);
  WgradArgsB a{(const __bf16 *)A, lda, (const __bf16 *)B, ldb, Me, Na, Nb, (float *)workspace, p.S, p.rows_per_slice,
               Nap, Nbp, (const __bf16 *)flags, (const __bf16 *)zrow, colsum ? 1 : 0, rowmap};
  const int ntiles = (Nap / p.T) * (Nbp / p.T);
  const unsigned grid = (unsigned)(((p.S + 7) / 8 * 8) * ntiles);
  if (p.T == 256 && nodes) proj_wgrad_bf16_kernel<256, true><<<grid, 512, 0, st>>>(a);
  else if (p.T == 256) proj_wgrad_bf16_kernel<256, false><<<grid, 512, 0, st>>>(a);
  else if (nodes) proj_wgrad_bf16_kernel<128, true><<<grid, 256, 0, st>>>(a);
  else proj_wgrad_bf16_kernel<128, false><<<grid, 256, 0, st>>>(a);
  wgrad_reduce_bf16_kernel<<<(unsigned)((n_all / 4 + 31) / 32), 256, 0, st>>>(
      (const float *)workspace, p.S, Na, Nb, Nap, Nbp, (unsigned short *)dW, (unsigned short *)colsum);
  return ampconv_launch_status();
}
