// Stream-K MFMA GEMM for the "NT" orientation: C[M x N] = alpha * rs[m] * cs[n] * A B^T + beta * C with BOTH operands
// contiguous along the contraction axis (A(m,k) = A[m*lda + k], B(k,n) = B[n*ldb + k]).  This is the covariance / Gram
// product of the path - X X^T over the time-major field (xmca/array.py:474-483 on the dual side, :553-566) - and the
// orientation the general kernel of gemm.h handles worst (transposing scalar LDS stores, 64 % of its LDS cycles were
// bank conflicts; split-K through an f64 workspace of splits x M x N).  Included by gemm.h.
//
// * LDS tiles keep the global orientation, [row][k] with a row pitch of BK + 2 (f64) / BK + 4 (f32) elements: the
//   16-byte global vectors go to LDS as they are (ds_write_b128), and the MFMA operand reads A[i = lane & 15][k = lane >> 4]
//   are conflict-free for f64 (bank pair 4 r + 2 k over a 32-lane group) and 2-way for f32.
// * Stream-K schedule: the (tile, k-tile) units of the whole product are dealt in equal contiguous ranges to a grid of
//   resident workgroups (one per CU: 512 threads with up to 256 VGPRs each - the two-per-CU budget of 128 spills).  A range
//   covers the tail of one tile, whole tiles, and the head of another: at most TWO partial segments per workgroup go to a
//   workspace (<= grid x 2 x 128 KB in total, instead of splits x M x N), whole tiles are written directly, and a second
//   small launch sums the segments of every split tile in workgroup order (deterministic) and applies the epilogue.
//   No quantisation loss in the last round of tiles, no split-count heuristic.
// * double-buffered LDS, one barrier per k-tile; the global loads of k-tile t + 2 are in flight while t is multiplied.
//   f32 operands: v_mfma_f32_16x16x4_f32 with the accumulators flushed into f64 every GEMM_FLUSH_TILES k-tiles (as in
//   gemm.h); f64: v_mfma_f64_16x16x4_f64.
#pragma once

namespace xmca {

template <typename TI>
struct NtCfg;
#ifndef XMCA_NT_F64_WGS
#define XMCA_NT_F64_WGS 1
#endif
#ifndef XMCA_NT_F32_WGS
#define XMCA_NT_F32_WGS 2
#endif
template <>
struct NtCfg<double> { static constexpr int PITCH = GEMM_BK + 2, WGS_PER_CU = XMCA_NT_F64_WGS, NBUF = 3 - WGS_PER_CU, MINW = 2 * WGS_PER_CU; };
template <>
struct NtCfg<float> { static constexpr int PITCH = GEMM_BK + 4, WGS_PER_CU = XMCA_NT_F32_WGS, NBUF = 3 - WGS_PER_CU, MINW = 2 * WGS_PER_CU; };

template <typename TI, typename TO>
struct GemmNtParams {
  const TI* A;
  const TI* B;
  TO* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  double alpha, beta;
  const double* row_scale;
  const double* col_scale;
  int upper_only, mirror;
  int n_tiles;
  const int* tile_map;      // packed (bm << 16 | bn), super-block order
  int nkt;                  // k-tiles per output tile
  long long total_units;    // n_tiles * nkt
  int units_per_wg;
  double* partial;          // [grid][2][GEMM_BM * GEMM_BN] raw sums of partial segments
  int vec_a, vec_b;
  int slabs;                // > 0: tile x K-slab schedule (one segment per workgroup, partial slot = slab * n_tiles + tile)
};

// WGS: workgroups per CU the build is sized for (1: 256 registers per lane and two LDS buffers; 2: 128 registers, one buffer)
template <typename TI, typename TO, bool WIDE, int WGS>
__global__ __launch_bounds__(GEMM_THREADS, 2 * WGS) void gemm_nt_kernel(GemmNtParams<TI, TO> p) {
  using M_ = Mfma<TI>;
  using acc_t = typename M_::acc_t;
  using vec_t = typename M_::vec_t;
  using IO = GemmTileIO<TI, true>;
  constexpr int PITCH = NtCfg<TI>::PITCH, NBUF = 3 - WGS, VW = M_::VW, NV = 8 / VW;
  __shared__ __attribute__((aligned(16))) TI As[NBUF][GEMM_BM][PITCH];
  __shared__ __attribute__((aligned(16))) TI Bs[NBUF][GEMM_BN][PITCH];

  // XCD-aware, bijective remap of the launch index: the workgroups of one XCD get neighbouring unit ranges
  int L;
  {
    const int b = blockIdx.x, n = gridDim.x, q = n / 8, r = n % 8, xcd = b % 8, idx = b / 8;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 2) * 64, wn = (wave & 3) * 32;
  const int l15 = lane & 15, l4 = lane >> 4;

  auto store_tile = [&](TI (*S)[PITCH], const TI (&reg)[8]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int r, k;
      IO::pos(i, r, k);
      vec_t x;
#pragma unroll
      for (int j = 0; j < VW; ++j) x[j] = reg[i * VW + j];
      *reinterpret_cast<vec_t*>(&S[r][k]) = x;
    }
  };

  // two schedules: stream-K (equal contiguous unit ranges) or, `slabs` > 0, tile x K-slab: workgroup b owns tile b % n_tiles
  // over the slab b / n_tiles of the contraction - the workgroups of one slab walk through the same columns of the operands
  // at the same time, so every panel is fetched from HBM once and re-read from L2 / Infinity Cache by the other tiles
  long long u, uend;
  if (p.slabs > 0) {
    const int tile = (int)blockIdx.x % p.n_tiles, sl = (int)blockIdx.x / p.n_tiles;
    L = sl * p.n_tiles + tile;
    const int k0 = (int)((long long)sl * p.nkt / p.slabs), k1 = (int)((long long)(sl + 1) * p.nkt / p.slabs);
    u = (long long)tile * p.nkt + k0;
    uend = (long long)tile * p.nkt + k1;
  } else {
    u = (long long)L * p.units_per_wg;
    uend = u + p.units_per_wg < p.total_units ? u + p.units_per_wg : p.total_units;
  }
  int seg = 0;
  while (u < uend) {
    const int tile = (int)(u / p.nkt), kt0 = (int)(u - (long long)tile * p.nkt);
    const int kt1 = (long long)(p.nkt - kt0) < uend - u ? p.nkt : kt0 + (int)(uend - u);
    const int packed = p.tile_map[tile];
    const int bm = packed >> 16, bn = packed & 0xffff;
    const int bm0 = bm * GEMM_BM, bn0 = bn * GEMM_BN;
    const int kbeg = kt0 * GEMM_BK, kend = min(p.K, kt1 * GEMM_BK);

    acc_t acc[4][2];
    d4_t wide[WIDE ? 4 : 1][WIDE ? 2 : 1];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = acc_t{0, 0, 0, 0};
        if constexpr (WIDE) wide[i][j] = d4_t{0, 0, 0, 0};
      }

    const bool interior = (bm0 + GEMM_BM <= p.M) && (bn0 + GEMM_BN <= p.N) && p.vec_a && p.vec_b;
    const TI* __restrict__ baseA = p.A + (int64_t)bm0 * p.lda;
    const TI* __restrict__ baseB = p.B + (int64_t)bn0 * p.ldb;
    TI ra[8], rb[8];
    auto load = [&](int k0) {
      if (interior && k0 + GEMM_BK <= kend) {
        IO::load_fast(baseA + k0, p.lda, ra);
        IO::load_fast(baseB + k0, p.ldb, rb);
      } else {
        IO::load_slow(p.A, p.lda, bm0, p.M, k0, kend, ra);
        IO::load_slow(p.B, p.ldb, bn0, p.N, k0, kend, rb);
      }
    };
    const int nkt = kt1 - kt0;
    load(kbeg);
    if constexpr (NBUF == 2) {
      store_tile(As[0], ra);
      store_tile(Bs[0], rb);
      __syncthreads();
      if (nkt > 1) load(kbeg + GEMM_BK);
    }
    for (int kt = 0; kt < nkt; ++kt) {
      const int cur = NBUF == 2 ? (kt & 1) : 0;
      if constexpr (NBUF == 1) {
        store_tile(As[0], ra);
        store_tile(Bs[0], rb);
        __syncthreads();
        if (kt + 1 < nkt) load(kbeg + (kt + 1) * GEMM_BK);
      }
#pragma unroll
      for (int k4 = 0; k4 < GEMM_BK / 4; ++k4) {
        const int kc = k4 * 4 + l4;
        TI a[4], b[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = As[cur][wm + i * 16 + l15][kc];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = Bs[cur][wn + j * 16 + l15][kc];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = M_::mma(a[i], b[j], acc[i][j]);
      }
      if constexpr (WIDE) {
        if ((kt % GEMM_FLUSH_TILES) == GEMM_FLUSH_TILES - 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
              for (int r = 0; r < 4; ++r) wide[i][j][r] += (double)acc[i][j][r];
              acc[i][j] = acc_t{0, 0, 0, 0};
            }
        }
      }
      if constexpr (NBUF == 2) {
        // the other buffer was last read in iteration kt - 1, which every wave left through the barrier below
        if (kt + 1 < nkt) {
          store_tile(As[cur ^ 1], ra);
          store_tile(Bs[cur ^ 1], rb);
        }
        __syncthreads();
        if (kt + 2 < nkt) load(kbeg + (kt + 2) * GEMM_BK);
      } else {
        __syncthreads();   // every wave is done with this k-tile before it is overwritten
      }
    }

    const bool whole = (kt0 == 0) && (kt1 == p.nkt);
    if (whole) {
      const bool offdiag = (bm != bn);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = bm0 + wm + i * 16 + M_::row(lane, r);
            const int col = bn0 + wn + j * 16 + l15;
            if (row < p.M && col < p.N) {
              double v = (double)acc[i][j][r];
              if constexpr (WIDE) v += wide[i][j][r];
              v *= p.alpha;
              if (p.row_scale) v *= p.row_scale[row];
              if (p.col_scale) v *= p.col_scale[col];
              const int64_t o = (int64_t)row * p.ldc + col;
              if (p.beta != 0.0) v += p.beta * (double)p.C[o];
              p.C[o] = (TO)v;
              if (p.mirror != 0 && offdiag) p.C[(int64_t)col * p.ldc + row] = (TO)(p.mirror > 0 ? v : -v);
            }
          }
    } else {
      // raw partial sums of this segment: slot 0 = the first segment of this workgroup, slot 1 = a later one (only the
      // last segment of a range can be partial besides the first)
      double* __restrict__ W = p.partial + (p.slabs > 0 ? (size_t)L : (size_t)L * 2 + (seg == 0 ? 0 : 1)) * (size_t)(GEMM_BM * GEMM_BN);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double v = (double)acc[i][j][r];
            if constexpr (WIDE) v += wide[i][j][r];
            W[(wm + i * 16 + M_::row(lane, r)) * GEMM_BN + wn + j * 16 + l15] = v;
          }
    }
    u += nkt;
    ++seg;
  }
}

// sums the partial segments of every split tile (in workgroup order) and applies the epilogue of the GEMM
template <typename TO>
__global__ __launch_bounds__(256) void gemm_nt_fixup_kernel(const double* __restrict__ partial, const int* __restrict__ split_tiles,
                                                            const int* __restrict__ tile_map, int nkt, int units_per_wg, TO* __restrict__ C,
                                                            int M, int N, int64_t ldc, double alpha, double beta,
                                                            const double* __restrict__ row_scale, const double* __restrict__ col_scale,
                                                            int mirror) {
  const int tile = split_tiles[blockIdx.x];
  const int packed = tile_map[tile];
  const int bm = packed >> 16, bn = packed & 0xffff;
  const long long u0 = (long long)tile * nkt, u1 = u0 + nkt - 1;
  const int w0 = (int)(u0 / units_per_wg), w1 = (int)(u1 / units_per_wg);
  constexpr int TE = GEMM_BM * GEMM_BN;
  for (int e = threadIdx.x; e < TE; e += 256) {
    double v = 0.0;
    for (int w = w0; w <= w1; ++w) {
      // slot 0 = the first segment of workgroup w, i.e. its range starts inside this tile; otherwise the tile is the
      // last segment of a range that began in an earlier tile (slot 1)
      const int slot = ((long long)w * units_per_wg / nkt) == tile ? 0 : 1;
      v += partial[((size_t)w * 2 + slot) * TE + e];
    }
    const int row = bm * GEMM_BM + e / GEMM_BN, col = bn * GEMM_BN + e % GEMM_BN;
    if (row < M && col < N) {
      v *= alpha;
      if (row_scale) v *= row_scale[row];
      if (col_scale) v *= col_scale[col];
      const int64_t o = (int64_t)row * ldc + col;
      if (beta != 0.0) v += beta * (double)C[o];
      C[o] = (TO)v;
      if (mirror != 0 && bm != bn) C[(int64_t)col * ldc + row] = (TO)(mirror > 0 ? v : -v);
    }
  }
}

// tile x K-slab schedule: partial[slab][tile] summed over the slabs in order, then the epilogue
template <typename TO>
__global__ __launch_bounds__(256) void gemm_nt_slab_fixup_kernel(const double* __restrict__ partial, const int* __restrict__ tile_map,
                                                                 int n_tiles, int slabs, TO* __restrict__ C, int M, int N, int64_t ldc,
                                                                 double alpha, double beta, const double* __restrict__ row_scale,
                                                                 const double* __restrict__ col_scale, int mirror) {
  const int tile = blockIdx.x;
  const int packed = tile_map[tile];
  const int bm = packed >> 16, bn = packed & 0xffff;
  constexpr int TE = GEMM_BM * GEMM_BN;
  for (int e = threadIdx.x; e < TE; e += 256) {
    double v = 0.0;
    for (int sl = 0; sl < slabs; ++sl) v += partial[((size_t)sl * n_tiles + tile) * TE + e];
    const int row = bm * GEMM_BM + e / GEMM_BN, col = bn * GEMM_BN + e % GEMM_BN;
    if (row < M && col < N) {
      v *= alpha;
      if (row_scale) v *= row_scale[row];
      if (col_scale) v *= col_scale[col];
      const int64_t o = (int64_t)row * ldc + col;
      if (beta != 0.0) v += beta * (double)C[o];
      C[o] = (TO)v;
      if (mirror != 0 && bm != bn) C[(int64_t)col * ldc + row] = (TO)(mirror > 0 ? v : -v);
    }
  }
}

// returns false when the problem should go to the general kernel instead (disabled / tiny)
template <typename TI, typename TO>
bool gemm_nt(hipStream_t st, GemmWorkspace& ws, GemmNtWorkspace& nws, const TI* A, int64_t lda, const TI* B, int64_t ldb, TO* C,
             int64_t ldc, int M, int N, int K, const GemmOpts& o) {
  // Measured on MI355X (scripts/gemm_nt_check.py, Gram products of the path): float32 fields, two workgroups per CU - as
  // fast as the general kernel (C5 20.1 vs 20.0 ms, C3 5.4 vs 5.5 ms) with a tenth of its partial-sum traffic (C5: 0.13
  // instead of 1.47 GB) -> default for float32.  float64: the two-per-CU build spills (292 bytes per lane, 2.5 vs 1.9 ms
  // at C2) and one workgroup per CU does not cover its own barriers (2.2 ms) -> general kernel unless XMCA_GEMM_NT=1.
  // Negative results kept out of the tree: a 256 x 128 block tile (one workgroup per CU, 256 VGPRs) was 0-25 % slower
  // than the 128 x 128 kernel on every shape of the path; the field stored transposed (TN) gains 3-7 %.
  static const int mode = [] { const char* e = std::getenv("XMCA_GEMM_NT"); return e ? std::atoi(e) : -1; }();   // -1: float32 only
  const bool enabled = mode == 1 || (mode == -1 && std::is_same<TI, float>::value);
  if (!enabled || K <= 0 || o.force_splits > 0) return false;
  if (nws.n_cus == 0) {
    int dev = 0, n = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    nws.n_cus = n > 0 ? n : 256;
  }
  const int tm = ceil_div(M, GEMM_BM), tn = ceil_div(N, GEMM_BN);
  const int64_t tiles = o.upper_only ? (int64_t)tm * (tm + 1) / 2 : (int64_t)tm * tn;
  const int nkt = ceil_div(K, GEMM_BK);
  const GemmWorkspace::Map& map = ws.tile_map(st, tm, tn, o.upper_only);
  XMCA_CHECK(map.n == tiles && tm < 65536 && tn < 65536, XMCA_ERR_INVALID, "gemm: tile map mismatch");
  constexpr bool WIDE = std::is_same<TI, float>::value;
  constexpr int VW = Mfma<TI>::VW;
  const long long total = (long long)tiles * nkt;
  const int vec_a = (lda % VW == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0);
  const int vec_b = (ldb % VW == 0) && (reinterpret_cast<uintptr_t>(B) % 16 == 0);
  // Few output tiles and a long contraction (the Gram matrix of a field with many more columns than rows: C5 has 55 upper
  // tiles and 32 400 k-tiles): tile x K-slab schedule, one workgroup per CU with the 256-register build (no spills) and
  // double-buffered LDS.  Stream-K gives every tile's K-range to neighbouring workgroups, which then share nothing: 58 GB of
  // operand traffic at C5 for a 5 GB field; here the workgroups of a slab read the same panels at the same time.
  static const bool slab_on = [] { const char* e = std::getenv("XMCA_NT_SLABS"); return !(e && e[0] == '0'); }();
  static const int slab_wgs = [] { const char* e = std::getenv("XMCA_NT_SLAB_WGS"); return e ? std::atoi(e) : 2; }();
  const int slots = nws.n_cus * (slab_wgs == 1 ? 1 : 2);
  if (slab_on && tiles <= slots / 2 && (long long)nkt >= 64LL * (slots / tiles)) {
    static const int slabs_forced = [] { const char* e = std::getenv("XMCA_NT_SLAB_COUNT"); return e ? std::atoi(e) : 0; }();
    const int slabs = slabs_forced > 0 ? slabs_forced : (int)(slots / tiles);
    double* W = nws.partial.ensure((size_t)slabs * tiles * GEMM_BM * GEMM_BN);
    GemmNtParams<TI, TO> p{A, B, C, M, N, K, lda, ldb, ldc, o.alpha, o.beta, o.row_scale, o.col_scale, o.upper_only ? 1 : 0, o.mirror,
                           (int)tiles, map.dev.get(), nkt, total, nkt, W, vec_a, vec_b, slabs};
    if (o.ev_begin) XMCA_HIP(hipEventRecord(o.ev_begin, st));
    if (slab_wgs == 1) hipLaunchKernelGGL((gemm_nt_kernel<TI, TO, WIDE, 1>), dim3((unsigned)(slabs * tiles)), dim3(GEMM_THREADS), 0, st, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<TI, TO, WIDE, 2>), dim3((unsigned)(slabs * tiles)), dim3(GEMM_THREADS), 0, st, p);
    if (o.ev_end) XMCA_HIP(hipEventRecord(o.ev_end, st));
    hipLaunchKernelGGL((gemm_nt_slab_fixup_kernel<TO>), dim3((unsigned)tiles), dim3(256), 0, st, W, map.dev.get(), (int)tiles, slabs, C, M, N, ldc,
                       o.alpha, o.beta, o.row_scale, o.col_scale, o.mirror);
    XMCA_HIP(hipGetLastError());
    return true;
  }
  const int max_wgs = nws.n_cus * NtCfg<TI>::WGS_PER_CU;
  // ranges: whole tiles when there are plenty of them (no partial segments at all), equal unit ranges otherwise
  int n_wg, upw;
  if (tiles >= 8LL * max_wgs) {
    n_wg = max_wgs;
    upw = (int)((tiles + n_wg - 1) / n_wg) * nkt;
  } else {
    n_wg = (int)std::min<long long>(max_wgs, std::max<long long>(1, total / 4));
    upw = (int)((total + n_wg - 1) / n_wg);
  }
  n_wg = (int)((total + upw - 1) / upw);
  // split tiles: every tile whose units are not inside one range (the list is kept until the schedule changes)
  if (nws.key_tiles != tiles || nws.key_nkt != nkt || nws.key_upw != upw) {
    std::vector<int> split;
    if (upw % nkt != 0)
      for (int t = 0; t < tiles; ++t)
        if ((long long)t * nkt / upw != ((long long)(t + 1) * nkt - 1) / upw) split.push_back(t);
    if (!split.empty()) {
      XMCA_HIP(hipMemcpyAsync(nws.split_tiles.ensure(split.size()), split.data(), sizeof(int) * split.size(), hipMemcpyHostToDevice, st));
      XMCA_HIP(hipStreamSynchronize(st));   // `split` is pageable host memory
    }
    nws.n_split = (int)split.size();
    nws.key_tiles = tiles; nws.key_nkt = nkt; nws.key_upw = upw;
  }
  double* W = nws.n_split > 0 ? nws.partial.ensure((size_t)n_wg * 2 * GEMM_BM * GEMM_BN) : nullptr;
  GemmNtParams<TI, TO> p{A, B, C, M, N, K, lda, ldb, ldc, o.alpha, o.beta, o.row_scale, o.col_scale, o.upper_only ? 1 : 0, o.mirror,
                         (int)tiles, map.dev.get(), nkt, total, upw, W, vec_a, vec_b, 0};
  if (o.ev_begin) XMCA_HIP(hipEventRecord(o.ev_begin, st));
  hipLaunchKernelGGL((gemm_nt_kernel<TI, TO, WIDE, NtCfg<TI>::WGS_PER_CU>), dim3(n_wg), dim3(GEMM_THREADS), 0, st, p);
  if (o.ev_end) XMCA_HIP(hipEventRecord(o.ev_end, st));
  XMCA_HIP(hipGetLastError());
  if (nws.n_split > 0) {
    hipLaunchKernelGGL((gemm_nt_fixup_kernel<TO>), dim3((unsigned)nws.n_split), dim3(256), 0, st, W, nws.split_tiles.get(), map.dev.get(), nkt,
                       upw, C, M, N, ldc, o.alpha, o.beta, o.row_scale, o.col_scale, o.mirror);
    XMCA_HIP(hipGetLastError());
  }
  return true;
}

}  // namespace xmca
