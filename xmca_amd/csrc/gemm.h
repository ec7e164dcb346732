// MFMA GEMM for gfx950: C[MxN] = alpha * rs[m] * cs[n] * op(A) * op(B) + beta * C          (round 4: rewritten)
//
// * f64 operands -> v_mfma_f64_16x16x4_f64, f32 operands -> v_mfma_f32_16x16x4_f32 (exact f32) with three levels of
//   accumulation ("WIDE", see GEMM_FLUSH_PRODUCTS) - or, for float32 products with both operands contiguous along the
//   contraction - or A so and B row-fast, the back-projection - six v_mfma_f32_32x32x16_bf16 on three-way bfloat16 pieces of
//   the operands ("X3").
// * 128 x 128 block tile, 256 threads = 4 waves (2 x 2), each wave a 64 x 64 tile = 4 x 4 MFMA tiles; two workgroups per CU
//   (one wave of each on every SIMD, 256 registers per lane each: accumulators + second-level sums + two sets of fragments
//   fit, no scratch in any instantiation).
// * operands go global -> LDS by LDS-DMA (`global_load_lds`, 16 bytes per lane, no staging registers) into two stages of
//   16 KB per operand (BK = 16 doubles / 32 floats = 128 bytes per row).  ONE barrier per k-tile, placed before the LAST
//   phase of a tile: behind it the DMA of k-tile t + 2 is issued into the stage everybody has finished reading, the first
//   fragments of k-tile t + 1 are requested, and the MFMAs of the last phase of t cover both.
// * fragments are read one phase (8 contraction indices) ahead into a second register set, so no MFMA waits for LDS.
// * LDS keeps the orientation of the global array, so nothing is transposed on the way in:
//     K-fast operand  (A(m,k) = A[m*lda + k], B(k,n) = B[n*ldb + k]):  [row][BK], 128-byte rows, 16-byte chunks XOR-swizzled
//         by (row >> 1) & 7 on the SOURCE address (the DMA writes lane-linear) and on the read;
//     row-fast operand (A(m,k) = A[k*lda + m], B(k,n) = B[k*ldb + n]): [k][128]; f32: chunk bit 3 flipped on odd k-row pairs.
//   Every fragment read fetches TWO elements per lane (ds_read_b128 for f64, ds_read_b64 for f32), conflict-free, and feeds
//   two MFMAs: for a K-fast operand two consecutive k of one row (k = 2 * (lane >> 4) + step), for a row-fast operand two
//   consecutive rows at one k, i.e. two interleaved MFMA tiles (the epilogue knows the interleaving).  Both use the same
//   k <-> (step, lane) map, so all four orientation pairs share one kernel body.
// * upper_only: only block tiles with bn >= bm are launched (Gram / Hermitian products); mirror = +1/-1 writes the
//   (anti)symmetric counterpart of off-diagonal tiles.
// * split-K inside the launch: every slice writes its raw f64 sums in register order (write-through 16-byte stores, 1 KB per
//   instruction) to a workspace slab, drains them, takes a ticket on the tile's counter, and the LAST arriver (one
//   agent-scope acquire) adds the slabs in slice order (bit-reproducible whatever the arrival order), applies the epilogue
//   and resets the counter (cdna_hip_programming.md 5, "in-launch split-K reduction", sc1 form).  No reduce kernel.
// * the (slice, tile) list is dealt to the 8 XCDs in contiguous ranges (workgroup b runs on XCD b % 8) and tiles are
//   enumerated in 8 x 8 super-blocks, so the workgroups sharing one L2 work on neighbouring tiles of the same k-slice.
// * the LDS-DMA needs no alignment beyond the element's (odd leading dimensions, odd bases: same rate); only the last,
//   partial k-tile (and operands whose tile offsets do not fit 32 bits) is staged through registers with predicated element
//   loads (zero filled) into the same LDS image.
//
// Reference call sites this replaces: numpy `@` / gesdd inner products of xmca/array.py:479, :552-566 and :580-584
// (see DESIGN.md for the formulation).
#pragma once
#include <algorithm>
#include <cmath>
#include <memory>
#include <type_traits>
#include <vector>

#include "common.h"

namespace xmca {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));

template <typename T>
struct Mfma;
template <>
struct Mfma<double> {
  using acc_t = d4_t;
  using vec_t = d2_t;      // one 16-byte chunk
  using frag_t = d2_t;     // one fragment read: two elements
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct Mfma<float> {
  using acc_t = f4_t;
  using vec_t = f4_t;
  using frag_t = f2_t;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_THREADS = 256;
constexpr int GEMM_STAGE_BYTES = 16384;     // one operand, one stage: 128 rows x 128 bytes
// WIDE (f32 operands): three levels of accumulation.  The MFMA accumulators take at most 512 products, are then added
// (f32) into a second set of f32 sums, which takes at most 32 such blocks = 16 384 products = one k-slice; longer
// contractions are ALWAYS cut into slices of at most that length, and slices are added in f64 (slabs, last arriver).
// Rounding: 512 products in f32 ~ sqrt(512) eps/2 relative to the block sum (random walk; 1.3e-6), 32 block sums another
// sqrt(32) eps/2 (3.4e-7), averaged over the 32 blocks: ~4e-7 per slice, ~4e-7 / sqrt(slices) for the whole contraction -
// 5e-8 at K = 1 036 800 (C5) - against eps/2 = 6e-8 of each f32 input element.  Round 1-3 flushed into f64 side
// accumulators instead (3e-8 at C5): 128 more registers per lane, which left no room for the second fragment set.
constexpr int GEMM_FLUSH_PRODUCTS = 512;
constexpr int GEMM_SLICE_PRODUCTS = 16384;
template <typename T>
struct GemmK {
  static constexpr int E = (int)sizeof(T);
  static constexpr int CE = 16 / E;                 // elements per 16-byte chunk
  static constexpr int BK = 128 / E;                // 16 doubles / 32 floats
  static constexpr int NPH = BK / 8;                // phases per k-tile: 8 contraction indices = 2 MFMA k-steps each
};

template <typename TI, typename TO>
struct GemmParams {
  const TI* A;
  const TI* B;
  TO* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  double alpha, beta;
  const double* row_scale;  // nullable, length M
  const double* col_scale;  // nullable, length N
  int upper_only;
  int mirror;
  int k_chunk;              // contraction length handled by one slice (multiple of BK)
  int splits;
  int n_tiles;              // block tiles per slice (upper triangle only when upper_only)
  int n_wg;                 // n_tiles * splits
  const int* tile_map;      // n_tiles packed (bm << 16 | bn) in super-block order
  const int* tile_k;        // nullable: per tile the contraction range [tile_k[2 t], tile_k[2 t + 1]) (GemmTileList), else [0, K)
  int vec_a, vec_b;         // LDS-DMA allowed (element-aligned base, tile offsets fit 32 bits)
  double* slabs;            // split-K: [splits][n_tiles][128 * 128] raw sums in register order
  int* counters;            // split-K: arrival tickets, one per tile, zero between launches
};

// One operand of the product: where its 16-byte chunks sit in a 16 KB LDS stage, how they are fetched, and which element
// of the 64-wide wave tile an MFMA operand lane holds.
template <typename TI, bool KFAST>
struct GemmOperand {
  static constexpr int E = GemmK<TI>::E, CE = GemmK<TI>::CE, BK = GemmK<TI>::BK;
  using frag_t = typename Mfma<TI>::frag_t;
  // LDS byte offset o (multiple of 16) of a stage -> (row r of the 128-wide axis, first contraction index k) of the chunk
  static __device__ __forceinline__ void where(int o, int& r, int& k) {
    if constexpr (KFAST) {
      r = o >> 7;
      k = (((o >> 4) & 7) ^ ((r >> 1) & 7)) * CE;
    } else {
      k = o / (128 * E);
      int c = (o % (128 * E)) >> 4;
      if constexpr (E == 4) c ^= ((k >> 1) & 1) << 3;
      r = c * CE;
    }
  }
  // LDS-DMA instruction q = 4 * wave + j of a stage (fills the LDS bytes [1024 q, 1024 q + 1024), lane l the 16 bytes at
  // 1024 q + 16 l): byte offset of the lane's chunk relative to the tile origin at k0 = voff(lane, j & 1) + soff(wave, j);
  // the same map as where(), split into a lane part (two registers: even / odd j) and a scalar part
  static __device__ __forceinline__ uint32_t voff(int lane, int jpar, int64_t ld) {
    if constexpr (KFAST) return (uint32_t)((lane >> 3) * ld * E + (((lane & 7) ^ (((lane >> 4) + 4 * jpar) & 7)) << 4));
    else if constexpr (E == 8) return (uint32_t)(lane * 16);
    else return (uint32_t)((lane >> 5) * ld * E + (((lane & 31) ^ (jpar << 3)) << 4));
  }
  static __device__ __forceinline__ uint32_t soff(int wave, int j, int64_t ld) {
    const int q = 4 * wave + j;
    return (uint32_t)((KFAST ? 8 : (E == 8 ? 1 : 2)) * q * ld * E);
  }
  // bytes from the tile origin at k0 to the end of the operand (R rows of the 128-wide axis, K contraction indices): the
  // buffer descriptor's range.  Rows beyond the matrix come back as zeros instead of being read (their results are never
  // stored).  A row-fast chunk that straddles row R reads on into the next k-row - garbage for a row that is never stored
  // - which exists for every k-row but the last: see `ragged` in the kernel.
  static __device__ __forceinline__ uint32_t remaining(int R, int row0, int K, int k0, int64_t ld) {
    const int64_t n = KFAST ? ((int64_t)(R - 1 - row0) * ld + (K - k0)) * E : ((int64_t)(K - 1 - k0) * ld + (R - row0)) * E;
    return n > 0xffffffffll ? 0xffffffffu : (uint32_t)n;
  }
  // LDS byte address (within a stage) of fragment read f (0..3) of phase q for this lane = frag_addr(frag_base(w0, lane), q, f)
  // with q, f compile-time: one register per operand, the rest is immediates and one XOR.  w0 = 0 / 64: the wave's origin.
  //   K-fast:   f = 16-row block: two consecutive k (k = 8 q + 2 kk + {0, 1}) of row w0 + 16 f + i; the chunk index
  //             (8 q + 2 kk) * E / 16 = (4 q | kk) for f64, (2 q | kk >> 1) for f32 is XORed with the row's swizzle, and the
  //             phase bits are disjoint from the lane bits, so the phase enters as an XOR of the byte address
  //   row-fast: f = 2 * step + u: rows w0 + 32 u + 2 i, + 1 at k = 8 q + 2 kk + step; f32 rows (512 bytes) flip chunk bit 3
  //             (byte bit 7) on k-rows with (k >> 1) & 1 = kk & 1, which makes u an XOR as well
  static __device__ __forceinline__ int frag_base(int w0, int lane) {
    const int i = lane & 15, kk = lane >> 4;
    if constexpr (KFAST) {
      const int byteoff = 2 * kk * E;
      return (w0 + i) * 128 + (((byteoff >> 4) ^ ((i >> 1) & 7)) << 4) + (byteoff & 15);
    } else {
      const int g = (w0 >> 1) + i;               // pair index along the row axis (u = 0)
      if constexpr (E == 8) return 2 * kk * 1024 + g * 16;
      else return 2 * kk * 512 + (((g >> 1) ^ ((kk & 1) << 3)) << 4) + (g & 1) * 8;
    }
  }
  static __device__ __forceinline__ int frag_addr(int base, int q, int f) {
    if constexpr (KFAST) return (base ^ (q * 8 * E)) + f * 2048;
    else if constexpr (E == 8) return base + (8 * q + (f >> 1)) * 1024 + (f & 1) * 256;
    else return (base ^ ((f & 1) << 7)) + (8 * q + (f >> 1)) * 512;
  }
  // operand of MFMA tile t (0..3) at k-step e (0..1) out of the four fragments of a phase
  static __device__ __forceinline__ TI value(const frag_t (&f)[4], int e, int t) {
    if constexpr (KFAST) return f[t][e];
    else return f[2 * e + (t >> 1)][t & 1];
  }
  // position (0..63) inside the wave tile of index x16 (0..15) of MFMA tile t
  static __device__ __forceinline__ int index(int t, int x16) {
    if constexpr (KFAST) return 16 * t + x16;
    else return 32 * (t >> 1) + 2 * x16 + (t & 1);
  }
};

static __device__ __forceinline__ int gemm_lane_id() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// X3 (float operands, A contiguous along the contraction: the covariance / Gram product of a float32 field and its
// back-projection): every float32
// element is split into three bfloat16 pieces on the way from LDS to the matrix pipe - x = h + m + l, 8 mantissa bits each,
// see gemm_split3 - and the product runs as six
// v_mfma_f32_32x32x16_bf16 per tile and 16 contraction indices, a.b ~ hh + (hm + mh) + (mm + hl + lh); the three terms left
// out are below 2^-23 |a||b|, the rounding of the float32 inputs themselves.  bfloat16 MFMA is 16 x the float32 rate, so
// the six products cost 3/8 of the v_mfma_f32_16x16x4_f32 path; sums stay float32 in the accumulators with the same
// 512 / 16 384-product levels.  Same LDS image and staging as the float32 kernel.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f16v_t __attribute__((ext_vector_type(16)));
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));

// eight floats (k ascending) -> three vectors of eight bfloat16, x = h + m + l up to 2^-25 |x|: h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m), each rounded to nearest (v_cvt_pk_bf16_f32: two elements per instruction; the remainders x - h and
// x - h - m are exact).  Rounding rather than truncating makes the pieces of mixed sign, so the cross terms the product leaves
// out (m l, l m, l l: <= 2^-24 |a||b| each) are zero-mean and add up like rounding noise instead of a bias.  Per pair of
// elements: cvt_pk, two unpacks (shift / AND), one packed subtraction - twice - and a last cvt_pk: nine instructions.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ void gemm_split3(const f4_t& x0, const f4_t& x1, u4_t& h, u4_t& m, u4_t& l) {
#ifdef XMCA_X3_NOSPLIT_EXPERIMENT
  // (experiment, WRONG numbers: the three "pieces" are bit patterns of the raw floats - the k-loop without the VALU split, i.e. what
  //  pre-split bfloat16 planes could reach at best with this tile and pipeline; profiles/r06_bf16_presplit_gate.txt)
  h = __builtin_bit_cast(u4_t, x0);
  m = __builtin_bit_cast(u4_t, x1);
  l = h ^ m;
  return;
#endif
  const f2_t x[4] = {f2_t{x0[0], x0[1]}, f2_t{x0[2], x0[3]}, f2_t{x1[0], x1[1]}, f2_t{x1[2], x1[3]}};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned int hp = __builtin_bit_cast(unsigned int, __builtin_convertvector(x[q], bf16x2_t));
    h[q] = hp;
    const f2_t r = x[q] - f2_t{__uint_as_float(hp << 16), __uint_as_float(hp & 0xffff0000u)};
    const unsigned int mp = __builtin_bit_cast(unsigned int, __builtin_convertvector(r, bf16x2_t));
    m[q] = mp;
    const f2_t t = r - f2_t{__uint_as_float(mp << 16), __uint_as_float(mp & 0xffff0000u)};
    l[q] = __builtin_bit_cast(unsigned int, __builtin_convertvector(t, bf16x2_t));
  }
}

template <typename TI, typename TO, bool A_KFAST, bool B_NFAST, bool WIDE, bool X3 = false>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_kernel(GemmParams<TI, TO> p) {
  static_assert(!X3 || (std::is_same<TI, float>::value && A_KFAST && WIDE), "X3: float32, A contiguous along the contraction");
  using M_ = Mfma<TI>;
  using acc_t = typename M_::acc_t;
  using vec_t = typename M_::vec_t;
  using frag_t = typename M_::frag_t;
  using OA = GemmOperand<TI, A_KFAST>;
  using OB = GemmOperand<TI, !B_NFAST>;
  constexpr int CE = GemmK<TI>::CE, BK = GemmK<TI>::BK, NPH = GemmK<TI>::NPH, E = GemmK<TI>::E, SB = GEMM_STAGE_BYTES;
  // the ONLY LDS object of the kernel (a second one makes hipcc wait for every LDS-DMA before each ds_read):
  // stage s: A at s * 2 SB, B at s * 2 SB + SB; after the loop the first word doubles as the "last arriver" flag
  __shared__ __attribute__((aligned(1024))) char smem[4 * SB];

  // XCD-aware, bijective remap of the launch index (cdna_hip_programming.md T1)
  int L;
  {
    const int b = blockIdx.x, q = p.n_wg / 8, r = p.n_wg % 8, xcd = b % 8, idx = b / 8;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int split = L / p.n_tiles, tile = L % p.n_tiles;
  const int packed = p.tile_map[tile];
  const int bm = packed >> 16, bn = packed & 0xffff;
  const int bm0 = bm * GEMM_BM, bn0 = bn * GEMM_BN;
  int kbeg, kend;
  if (p.tile_k) {             // block-sparse products: the tile's own contraction range, cut into p.splits slices
    const int kb = p.tile_k[2 * tile], ke = p.tile_k[2 * tile + 1];
    const int kc = ((ke - kb + GemmK<TI>::BK * p.splits - 1) / (GemmK<TI>::BK * p.splits)) * GemmK<TI>::BK;
    kbeg = kb + split * kc;
    kend = max(kbeg, min(ke, kbeg + kc));
  } else {
    kbeg = split * p.k_chunk;
    kend = min(p.K, kbeg + p.k_chunk);
  }

  // Nothing derived from the lane index has to stay in a register across the main loop except the two fragment addresses
  // and the four DMA offsets: the f32 build has 64 + 128 accumulator registers and no room for bystanders, so the lane
  // index is produced again (v_mbcnt, opaque to the optimiser) behind the loop.
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  int lane = gemm_lane_id();
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  acc_t acc[X3 ? 1 : 4][X3 ? 1 : 4];
  acc_t wide[(WIDE && !X3) ? 4 : 1][(WIDE && !X3) ? 4 : 1];      // second-level f32 sums (see GEMM_FLUSH_PRODUCTS)
  f16v_t acc3[X3 ? 2 : 1][X3 ? 2 : 1], wide3[X3 ? 2 : 1][X3 ? 2 : 1];   // X3: 2 x 2 tiles of 32 x 32
  if constexpr (X3) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 16; ++g) { acc3[i][j][g] = 0.f; wide3[i][j][g] = 0.f; }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = acc_t{0, 0, 0, 0};
        if constexpr (WIDE) wide[i][j] = acc_t{0, 0, 0, 0};
      }
  }

  // ---- staging ----
  // fast path (LDS-DMA): wave w issues the instructions q = 4 w + j (j = 0..3) of each operand; instruction q fills the
  // LDS bytes [q * 1024, q * 1024 + 1024) of the stage, lane l the 16 bytes at q * 1024 + 16 l
  const char* tileA = reinterpret_cast<const char*>(A_KFAST ? p.A + (int64_t)bm0 * p.lda : p.A + bm0);   // + k offset per tile
  const char* tileB = reinterpret_cast<const char*>(B_NFAST ? p.B + bn0 : p.B + (int64_t)bn0 * p.ldb);
  const int64_t kstepA = (A_KFAST ? 1 : p.lda) * E, kstepB = (B_NFAST ? p.ldb : 1) * E;
  uint32_t va[2], vb[2];
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) {
    va[jp] = OA::voff(lane, jp, p.lda);
    vb[jp] = OB::voff(lane, jp, p.ldb);
  }
  const bool fast = p.vec_a && p.vec_b;
  // buffer form of the LDS-DMA: descriptor base = tile origin + k offset and range = what is left of the operand
  // (scalars), 32-bit lane offsets, scalar offsets for the four instructions of a wave
  using lptr_t = __attribute__((address_space(3))) void*;
  auto stage_fast = [&](int k0, int s) {
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass has no buffer-descriptor type and would drop the kernel's launch stub)
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(tileA + (int64_t)k0 * kstepA), 0,
                                                                       (int)OA::remaining(p.M, bm0, p.K, k0, p.lda), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(tileB + (int64_t)k0 * kstepB), 0,
                                                                       (int)OB::remaining(p.N, bn0, p.K, k0, p.ldb), 0x00020000);
    char* la = smem + s * 2 * SB + wave * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(la + j * 1024), 16, va[j & 1], OA::soff(wave, j, p.lda), 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(la + SB + j * 1024), 16, vb[j & 1], OB::soff(wave, j, p.ldb), 0, 0);
#endif
  };
  // slow path: predicated element loads, zero filled, into the same image
  auto stage_slow = [&](int k0, int s, int tid) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = (tid + GEMM_THREADS * j) * 16;
      int r, k;
      vec_t x;
      OA::where(o, r, k);
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        const int er = bm0 + (A_KFAST ? r : r + e), ek = k0 + (A_KFAST ? k + e : k);
        TI v = TI(0);
        if (er < p.M && ek < kend) v = A_KFAST ? p.A[(int64_t)er * p.lda + ek] : p.A[(int64_t)ek * p.lda + er];
        x[e] = v;
      }
      *reinterpret_cast<vec_t*>(smem + s * 2 * SB + o) = x;
      OB::where(o, r, k);
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        const int er = bn0 + (!B_NFAST ? r : r + e), ek = k0 + (!B_NFAST ? k + e : k);
        TI v = TI(0);
        if (er < p.N && ek < kend) v = !B_NFAST ? p.B[(int64_t)er * p.ldb + ek] : p.B[(int64_t)ek * p.ldb + er];
        x[e] = v;
      }
      *reinterpret_cast<vec_t*>(smem + s * 2 * SB + SB + o) = x;
    }
  };
  // X3 fragments: lane (i = lane & 31, kk = lane >> 5) reads the 8 floats k = 16 q + 8 kk ... + 7 of row w0 + 32 t + i - two
  // neighbouring 16-byte chunks (c, c ^ 1 after the swizzle), conflict-free like the reads of the float32 path
  int fa0 = X3 ? (wm + (lane & 31)) * 128 + (((2 * (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4) : OA::frag_base(wm, lane);
  // (X3, B row-fast - the back-projection of a float32 field: the lane's 8 values are one float from each of the k-rows
  //  16 q + 8 kk + j at column w0 + 32 t + i: ds_read_b32 over 32 consecutive floats per lane group, conflict-free; row pairs
  //  (j, j + 1) share the row swizzle and sit 512 bytes apart - one ds_read2_b32)
  int fb0 = (X3 ? (B_NFAST ? (lane >> 5) * 4096 + (wn + (lane & 31)) * 4
                           : (wn + (lane & 31)) * 128 + (((2 * (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4))
                : OB::frag_base(wn, lane)) + SB;
  constexpr int NFR = X3 ? 1 : 4, NQ = X3 ? 2 : NPH;        // (X3: one phase = 16 contraction indices)
  frag_t af[X3 ? 1 : 2][NFR], bf[X3 ? 1 : 2][NFR];
  f4_t ar3[X3 ? 2 : 1][2], br3[X3 ? 2 : 1][2];              // X3: raw floats of the NEXT phase
  auto read_frags3 = [&](const char* sb, int q) {
    asm("" : "+v"(fa0), "+v"(fb0));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int oa = (fa0 ^ (q << 6)) + t * 4096;
      ar3[t][0] = *reinterpret_cast<const f4_t*>(sb + oa);
      ar3[t][1] = *reinterpret_cast<const f4_t*>(sb + (oa ^ 16));
      if constexpr (!B_NFAST) {
        const int ob = (fb0 ^ (q << 6)) + t * 4096;
        br3[t][0] = *reinterpret_cast<const f4_t*>(sb + ob);
        br3[t][1] = *reinterpret_cast<const f4_t*>(sb + (ob ^ 16));
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          br3[t][j >> 2][j & 3] = *reinterpret_cast<const float*>(sb + fb0 + (16 * q + j) * 512 + ((t ^ ((j >> 1) & 1)) << 7));
      }
    }
  };
  u4_t ah[X3 ? 2 : 1], am[X3 ? 2 : 1], al[X3 ? 2 : 1], bh[X3 ? 2 : 1], bm3[X3 ? 2 : 1], bl[X3 ? 2 : 1];
  auto split3 = [&]() {       // raw floats of this phase -> bfloat16 planes; ar3 / br3 are free for the next phase's reads afterwards
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      gemm_split3(ar3[t][0], ar3[t][1], ah[t], am[t], al[t]);
      gemm_split3(br3[t][0], br3[t][1], bh[t], bm3[t], bl[t]);
    }
  };
  auto mma3 = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
    auto B8 = [](const u4_t& v) { return __builtin_bit_cast(bf16x8_t, v); };
    // term by term over the four tiles: consecutive MFMAs never touch the same accumulator (small terms first)
#define XMCA_X3_TERM(P, Q)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
        acc3[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B8(P[i]), B8(Q[j]), acc3[i][j], 0, 0, 0);
    XMCA_X3_TERM(al, bh)
    XMCA_X3_TERM(ah, bl)
    XMCA_X3_TERM(am, bm3)
    XMCA_X3_TERM(am, bh)
    XMCA_X3_TERM(ah, bm3)
    XMCA_X3_TERM(ah, bh)
#undef XMCA_X3_TERM
#endif
  };
  auto read_frags = [&](const char* sb, int q, frag_t (&a)[NFR], frag_t (&b)[NFR]) {
    // (the per-phase XORs of the two base addresses are one instruction each; hoisted out of the loop they would occupy
    // up to six more registers for its whole duration - the empty asm makes the bases look freshly written)
#ifndef GEMM_NO_LAUNDER
    asm("" : "+v"(fa0), "+v"(fb0));
#endif
    if constexpr (!X3) {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        a[f] = *reinterpret_cast<const frag_t*>(sb + OA::frag_addr(fa0, q, f));
        b[f] = *reinterpret_cast<const frag_t*>(sb + OB::frag_addr(fb0, q, f));
      }
    }
  };
  auto mfmas = [&](const frag_t (&a)[NFR], const frag_t (&b)[NFR]) {
    if constexpr (!X3) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = M_::mma(OA::value(a, e, i), OB::value(b, e, j), acc[i][j]);
    }
  };

  const int nkt = (kend - kbeg + BK - 1) / BK;
  // complete k-tiles go through the pipelined LDS-DMA loop.  The DMA needs no alignment beyond the element's (measured:
  // odd leading dimensions and odd bases run at the same rate).  A row-fast operand whose row count is not a multiple of
  // the chunk has chunks that straddle its last row: in the LAST k-row of the matrix such a chunk would leave the operand,
  // so the k-tile holding that row takes the register path below.
  const bool ragged = (!A_KFAST && (p.M % CE) != 0) || (B_NFAST && (p.N % CE) != 0);
  int nfull = fast ? (kend - kbeg) / BK : 0;
  if (ragged && kend == p.K && nfull > 0 && nfull * BK == kend - kbeg) --nfull;
  constexpr int FLUSH_TILES = GEMM_FLUSH_PRODUCTS / BK;
  auto flush = [&](int kt) {
    if constexpr (X3) {
      if ((kt % FLUSH_TILES) == FLUSH_TILES - 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) { wide3[i][j][g] += acc3[i][j][g]; acc3[i][j][g] = 0.f; }
      }
    } else if constexpr (WIDE) {
      if ((kt % FLUSH_TILES) == FLUSH_TILES - 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) wide[i][j][r] += acc[i][j][r];
            acc[i][j] = acc_t{0, 0, 0, 0};
          }
      }
    }
  };
  if (nfull > 0) {
    stage_fast(kbeg, 0);
    __syncthreads();
    if (nfull > 1) stage_fast(kbeg + BK, 1);
    if constexpr (X3) read_frags3(smem, 0);
    else read_frags(smem, 0, af[0], bf[0]);
  }
  if constexpr (X3) {
    // the same pipeline with the split in front of the next phase's reads: split(q), reads(q + 1), six MFMAs per tile
    for (int kt = 0; kt < nfull; ++kt) {
      const char* sb = smem + (kt & 1) * 2 * SB;
      split3();
      read_frags3(sb, 1);
      mma3();
      split3();
      if (kt + 1 < nfull) {
        __syncthreads();
        if (kt + 2 < nfull) stage_fast(kbeg + (kt + 2) * BK, kt & 1);
        read_frags3(smem + ((kt + 1) & 1) * 2 * SB, 0);
      }
      mma3();
      flush(kt);
    }
  }
  for (int kt = 0; !X3 && kt < nfull; ++kt) {
    const char* sb = smem + (kt & 1) * 2 * SB;
#pragma unroll
    for (int q = 0; q < NPH; ++q) {
      if (q + 1 < NPH) {
        read_frags(sb, q + 1, af[(q + 1) & 1], bf[(q + 1) & 1]);
      } else if (kt + 1 < nfull) {
        // every read of k-tile kt has been issued; behind this barrier they have all completed in every wave and
        // k-tile kt + 1 has landed (vmcnt(0) of the DMA issued one tile ago)
        __syncthreads();
#if !defined(GEMM_ABLATE) || GEMM_ABLATE != 1
        if (kt + 2 < nfull) stage_fast(kbeg + (kt + 2) * BK, kt & 1);
#endif
        read_frags(smem + ((kt + 1) & 1) * 2 * SB, 0, af[0], bf[0]);
      }
      mfmas(af[q & 1], bf[q & 1]);
    }
    flush(kt);
  }
  lane = gemm_lane_id();
  const int tid = wave * 64 + lane;
  // the partial last k-tile, and every k-tile of unaligned operands: staged through registers, two barriers per tile
  for (int kt = nfull; kt < nkt; ++kt) {
    __syncthreads();
    stage_slow(kbeg + kt * BK, 0, tid);
    __syncthreads();
    if constexpr (X3) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        read_frags3(smem, q);
        split3();
        mma3();
      }
    } else {
#pragma unroll
      for (int q = 0; q < NPH; ++q) {
        read_frags(smem, q, af[0], bf[0]);
        mfmas(af[0], bf[0]);
      }
    }
    flush(kt);
  }

  // ---- totals of this slice as doubles, by accumulator register r = 0..63 of the lane ----
  //   native: r = (i * 4 + j) * 4 + q  (MFMA tile (i, j) of 16 x 16, register q);  X3: r = (i * 2 + j) * 16 + g  (32 x 32 tiles)
  d2_t tot[32];
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    double v;
    if constexpr (X3) v = (double)acc3[r >> 5][(r >> 4) & 1][r & 15] + (double)wide3[r >> 5][(r >> 4) & 1][r & 15];
    else {
      v = (double)acc[r >> 4][(r >> 2) & 3][r & 3];
      if constexpr (WIDE) v += (double)wide[r >> 4][(r >> 2) & 3][r & 3];
    }
    tot[r >> 1][r & 1] = v;
  }

  if (p.splits > 1) {
    // publish this slice's sums in register order - register pair h of thread tid at h * 256 + tid, 1 KB per store instruction -
    // with write-through (sc1) 16-byte stores: no release fence (the L2 write-back of a fence costs microseconds with
    // 128 KB freshly dirtied per workgroup: MI355X_MICROARCH.md "publish-large"), the drained stores + barrier + relaxed
    // ticket are the publication (cdna_hip_programming.md 5, sc1 form of the split-K hand-off)
    double* __restrict__ mine = p.slabs + ((size_t)split * p.n_tiles + tile) * (size_t)(GEMM_BM * GEMM_BN);
#if defined(__HIP_DEVICE_COMPILE__)
    {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(mine, 0, GEMM_BM * GEMM_BN * 8, 0x00020000);
#pragma unroll
      for (int h = 0; h < 32; ++h)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, tot[h]), rs, tid * 16, h * 4096, /*sc1*/ 16);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                      // also: every wave is done with the LDS stages
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) {
      const int ticket = __hip_atomic_fetch_add(p.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == p.splits - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(p.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      }
      *flag = last;
    }
    __syncthreads();
    if (*flag == 0) return;
    // the last arriver: slabs of all slices in slice order (its own comes back from memory too: same bits for every
    // arrival order)
#pragma unroll
    for (int h = 0; h < 32; ++h) tot[h] = d2_t{0, 0};
    for (int z = 0; z < p.splits; ++z) {
      const d2_t* __restrict__ sl = reinterpret_cast<const d2_t*>(p.slabs + ((size_t)z * p.n_tiles + tile) * (size_t)(GEMM_BM * GEMM_BN));
#pragma unroll
      for (int h0 = 0; h0 < 32; h0 += 8) {
#pragma unroll
        for (int h = h0; h < h0 + 8; ++h) tot[h] += sl[h * GEMM_THREADS + tid];
        asm volatile("" ::: "memory");     // eight loads in flight per lane, not thirty-two: the totals already take 128 registers
      }
    }
  }

  // ---- epilogue ----
  const bool offdiag = (bm != bn);
  TO* __restrict__ C = p.C;
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    int rin, cin;                          // position inside the 64 x 64 wave tile
    if constexpr (X3) {                    // C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (g & 3) + 8 (g >> 2) + 4 (lane >> 5)
      const int g = r & 15;
      rin = 32 * (r >> 5) + (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5);
      cin = 32 * ((r >> 4) & 1) + (lane & 31);
    } else {
      rin = OA::index(r >> 4, M_::row(lane, r & 3));
      cin = OB::index((r >> 2) & 3, lane & 15);
    }
    const int row = bm0 + wm + rin, col = bn0 + wn + cin;
    // symmetric results (mirror = +1): inside a DIAGONAL tile only the upper triangle is stored and mirrored as well - the two
    // halves are computed by different lanes, and in the X3 form with the three-way terms in a different order, so they agree
    // to rounding only; the matrix that leaves the kernel is symmetric bit for bit
    const bool sym_diag = p.mirror > 0 && !offdiag;
    if (row < p.M && col < p.N && !(sym_diag && col < row)) {
      double v = tot[r >> 1][r & 1] * p.alpha;
      if (p.row_scale) v *= p.row_scale[row];
      if (p.col_scale) v *= p.col_scale[col];
      const int64_t o = (int64_t)row * p.ldc + col;
      if (p.beta != 0.0) v += p.beta * (double)C[o];
      C[o] = (TO)v;
      if ((p.mirror != 0 && offdiag) || (sym_diag && col > row)) C[(int64_t)col * p.ldc + row] = (TO)(p.mirror > 0 ? v : -v);
    }
  }
}

// A caller-made list of block tiles, each with its own contraction range: block-sparse products (the compact-WY
// precomputation of tridiag_vec.h) as ONE launch.  K of the call stays the extent of the operands along the contraction.
struct GemmTileList {
  DevBuf<int> map;       // (bm << 16) | bn
  DevBuf<int> krange;    // [k0, k1) per tile, k0 a multiple of 16
  int n = 0;
  int klen = 0;          // longest range (split heuristic)
  void upload(hipStream_t st, const std::vector<int>& tiles, const std::vector<int>& kr) {
    n = (int)tiles.size();
    klen = 0;
    for (int t = 0; t < n; ++t) klen = std::max(klen, kr[2 * t + 1] - kr[2 * t]);
    if (n == 0) return;
    XMCA_HIP(hipMemcpyAsync(map.ensure(tiles.size()), tiles.data(), sizeof(int) * tiles.size(), hipMemcpyHostToDevice, st));
    XMCA_HIP(hipMemcpyAsync(krange.ensure(kr.size()), kr.data(), sizeof(int) * kr.size(), hipMemcpyHostToDevice, st));
    XMCA_HIP(hipStreamSynchronize(st));
  }
};

struct GemmOpts {
  bool a_kfast = true;   // A(m,k) = A[m*lda + k]
  bool b_nfast = true;   // B(k,n) = B[k*ldb + n]
  double alpha = 1.0, beta = 0.0;
  const double* row_scale = nullptr;
  const double* col_scale = nullptr;
  bool upper_only = false;
  int mirror = 0;
  int force_splits = 0;  // 0 = heuristic
  const GemmTileList* tiles = nullptr;   // block-sparse: only these tiles, each over its own contraction range (no upper_only / mirror)
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;   // optional: recorded around the MFMA kernel launch alone
};

// scratch for split-K slabs / tickets and the tile-order tables, owned by the caller (one per stream)
struct GemmWorkspace {
  DevBuf<double> slabs;
  DevBuf<int> counters;
  size_t counters_ready = 0;     // leading counters known to be zero
  struct Map { int tm, tn, upper; DevBuf<int> dev; int n; };
  std::vector<std::unique_ptr<Map>> maps;
  int n_cus = 0;

  // Tiles are enumerated in 8 x 8 super-blocks: the ~64 workgroups that share one XCD's L2 at a time then touch only
  // 8 + 8 operand panels, so each panel slice is fetched from HBM / Infinity Cache once per XCD and reused from L2.
  const Map& tile_map(hipStream_t st, int tm, int tn, bool upper) {
    for (auto& m : maps)
      if (m->tm == tm && m->tn == tn && m->upper == (int)upper) return *m;
    constexpr int G = 8;
    std::vector<int> order;
    for (int si = 0; si < tm; si += G)
      for (int sj = upper ? si : 0; sj < tn; sj += G)
        for (int i = si; i < std::min(si + G, tm); ++i)
          for (int j = sj; j < std::min(sj + G, tn); ++j)
            if (!upper || j >= i) order.push_back((i << 16) | j);
    auto m = std::make_unique<Map>();
    m->tm = tm; m->tn = tn; m->upper = upper; m->n = (int)order.size();
    XMCA_HIP(hipMemcpyAsync(m->dev.ensure(order.size()), order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice, st));
    XMCA_HIP(hipStreamSynchronize(st));
    maps.push_back(std::move(m));
    return *maps.back();
  }
  int* tickets(hipStream_t st, size_t n) {
    if (n > counters_ready) {
      const size_t want = std::max<size_t>(n, 4096);
      int* c = counters.ensure(want);
      XMCA_HIP(hipMemsetAsync(c, 0, sizeof(int) * counters.cap, st));
      counters_ready = counters.cap;
    }
    return counters.get();
  }
  int cus() {
    if (n_cus == 0) {
      int dev = 0, n = 256;
      (void)hipGetDevice(&dev);
      (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
      n_cus = n > 0 ? n : 256;
    }
    return n_cus;
  }
};

// float32 products whose A operand is contiguous along the contraction (Gram / covariance of a float32 field, its
// back-projection) by three-way bfloat16 splitting (X3 in the kernel): on unless XMCA_GEMM_BF16X3=0
static inline bool gemm_x3_enabled() {
  static const bool on = [] { const char* e = std::getenv("XMCA_GEMM_BF16X3"); return !(e && e[0] == '0'); }();
  return on;
}

template <typename TI, typename TO, bool WIDE>
static void launch_gemm_variant(hipStream_t st, const GemmParams<TI, TO>& p, bool a_kfast, bool b_nfast) {
  dim3 grid(p.n_wg), block(GEMM_THREADS);
  if constexpr (std::is_same<TI, float>::value) {
    if (a_kfast && gemm_x3_enabled()) {
      if (b_nfast) hipLaunchKernelGGL((gemm_kernel<TI, TO, true, true, WIDE, true>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((gemm_kernel<TI, TO, true, false, WIDE, true>), grid, block, 0, st, p);
      XMCA_HIP(hipGetLastError());
      return;
    }
  }
  if (a_kfast && b_nfast) hipLaunchKernelGGL((gemm_kernel<TI, TO, true, true, WIDE>), grid, block, 0, st, p);
  else if (a_kfast && !b_nfast) hipLaunchKernelGGL((gemm_kernel<TI, TO, true, false, WIDE>), grid, block, 0, st, p);
  else if (!a_kfast && b_nfast) hipLaunchKernelGGL((gemm_kernel<TI, TO, false, true, WIDE>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((gemm_kernel<TI, TO, false, false, WIDE>), grid, block, 0, st, p);
  XMCA_HIP(hipGetLastError());
}

// number of k-slices.  Cost model in units of one k-tile of a workgroup that has its CU to itself, fitted to sweeps of the
// split count on the shapes of the path (scripts/probes/gemm_probe.cpp sweep; profiles/r04_gemm_split_sweep.txt): a
// workgroup alone on a CU runs at ~0.75 of the matrix rate, two sharing one at ~0.87 together (1.73 units per k-tile
// each).  The workgroups of a launch all take the same time, so they run in rounds of 2 x CUs: full rounds at 1.73, a last
// round that fills at most half of the slots at 1 (one workgroup per CU).  8 units of prologue / epilogue / slab traffic
// per workgroup, one per slab the last arriver adds up, and half a percent per slice against ties.
static inline int gemm_choose_splits(int64_t tiles, int nkt, int n_cus) {
  const int max_s = std::min(std::max(nkt / 4, 1), 128);
  int best_s = 1;
  double best = 1e300;
  for (int s = 1; s <= max_s; ++s) {
    const double q = (double)tiles * s / (2.0 * n_cus);
    const double fl = std::floor(q), phi = q - fl;
    const double rounds = fl * 1.73 + (phi < 1e-9 ? 0.0 : (phi <= 0.5 ? 1.0 : 1.73));
    const double cost = ((double)nkt / s + 8.0) * rounds * (1.0 + 0.005 * s) + (s > 1 ? (double)s : 0.0);
    if (cost < best) { best = cost; best_s = s; }
  }
  return best_s;
}

// TI in {float,double}; TO in {float,double}.  f32 operands always use WIDE accumulation.
template <typename TI, typename TO>
void gemm(hipStream_t st, GemmWorkspace& ws, const TI* A, int64_t lda, const TI* B, int64_t ldb, TO* C, int64_t ldc, int M,
          int N, int K, const GemmOpts& o) {
  if (M <= 0 || N <= 0) return;
  XMCA_CHECK(!o.upper_only || M == N, XMCA_ERR_INVALID, "gemm: upper_only needs a square result");
  constexpr int BK = GemmK<TI>::BK, CE = GemmK<TI>::CE;
  const int tm = ceil_div(M, GEMM_BM), tn = ceil_div(N, GEMM_BN);
  constexpr bool WIDE = std::is_same<TI, float>::value;
  const GemmTileList* list = o.tiles;
  XMCA_CHECK(!list || (!o.upper_only && o.mirror == 0 && !WIDE), XMCA_ERR_INVALID, "gemm: tile lists are plain float64 products");
  if (list && list->n == 0) return;
  const int64_t tiles = list ? list->n : o.upper_only ? (int64_t)tm * (tm + 1) / 2 : (int64_t)tm * tn;
  const int klen = list ? list->klen : K;
  const int nkt = ceil_div(klen, BK);
  int splits = o.force_splits > 0 ? o.force_splits : gemm_choose_splits(tiles, nkt, ws.cus());
  if (WIDE) splits = std::max(splits, ceil_div(K, GEMM_SLICE_PRODUCTS));
  const int* map_dev = nullptr;
  if (list) {
    map_dev = list->map.get();
  } else {
    const GemmWorkspace::Map& map = ws.tile_map(st, tm, tn, o.upper_only);
    XMCA_CHECK(map.n == tiles && tm < 65536 && tn < 65536, XMCA_ERR_INVALID, "gemm: tile map mismatch");
    map_dev = map.dev.get();
  }
  // LDS-DMA: tile-relative byte offsets must fit 32 bits, elements naturally aligned (anything else: register path)
  const int vec_a = lda * (int64_t)sizeof(TI) * 128 < ((int64_t)1 << 32) && reinterpret_cast<uintptr_t>(A) % sizeof(TI) == 0;
  const int vec_b = ldb * (int64_t)sizeof(TI) * 128 < ((int64_t)1 << 32) && reinterpret_cast<uintptr_t>(B) % sizeof(TI) == 0;
  int k_chunk = klen > 0 ? klen : 1;
  if (splits > 1) {
    k_chunk = ceil_div(nkt, splits) * BK;
    if (WIDE) k_chunk = std::min(k_chunk, GEMM_SLICE_PRODUCTS);
    splits = ceil_div(klen, k_chunk);
  }
  if (splits < 1 || klen <= 0) splits = 1;
  XMCA_CHECK(tiles * splits < (int64_t)1 << 31, XMCA_ERR_INVALID, "gemm: too many workgroups");
  double* slabs = nullptr;
  int* counters = nullptr;
  if (splits > 1) {
    slabs = ws.slabs.ensure((size_t)splits * tiles * GEMM_BM * GEMM_BN);
    counters = ws.tickets(st, (size_t)tiles);
  }
  GemmParams<TI, TO> p{A, B, C, M, N, K, lda, ldb, ldc, o.alpha, o.beta, o.row_scale, o.col_scale, o.upper_only ? 1 : 0, o.mirror,
                       k_chunk, splits, (int)tiles, (int)(tiles * splits), map_dev, list ? list->krange.get() : nullptr, vec_a, vec_b,
                       slabs, counters};
  if (o.ev_begin) XMCA_HIP(hipEventRecord(o.ev_begin, st));
  launch_gemm_variant<TI, TO, WIDE>(st, p, o.a_kfast, o.b_nfast);
  if (o.ev_end) XMCA_HIP(hipEventRecord(o.ev_end, st));
}

}  // namespace xmca
