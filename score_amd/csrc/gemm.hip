// fp32 GEMM entry point (score_gemm) and the exact-fp32 kernel: v_mfma_f32_32x32x2_f32 (bitwise an
// fmaf chain in k order; gfx950 has no xf32/TF32).  Block = 2x2 waves of (32*WM)x(32*WN) MFMA tiles,
// BK = 16 (a 32-deep tile measured 6 % slower on this model's shapes) staged through LDS with a register
// prefetch of the next tile; NN / NT / TN operand layouts (see score_hip.h); split-K with a fixed-order
// slab reduce.  Large products are routed to gemm_bf16x3.hip when the caller allows it.  Also here: the
// deferred multi-job column sum (bias gradients, slab reductions), the grouped launches and the deferred
// weight-gradient queue.
#include <cstring>
#include "common.h"
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK_ALIGN 32   // split-K chunks are multiples of this
#ifndef GEMM_SPLITK_MIN_K
#define GEMM_SPLITK_MIN_K 512       // f32 kernel: split the reduced dimension only from this K on ...
#define GEMM_SPLITK_MIN_CHUNK 128   // ... and never into chunks shorter than this (tools/gemm_small_ab.py)
#endif

enum { F_BIAS = 1, F_RELU = 2, F_ACC = 4, F_DROP = 8, F_X3 = 16, F_X3F = 32, F_RELUGRAD = 64 };
#define KF_MASK (15 | 64 | 0x7FFF0000)   // what the kernels see: epilogue bits + the bias row group

// bias element of (row, col): one bias row, or one per group of g = flags >> 16 output rows
__device__ __forceinline__ int64_t bias_index(int flags, int row, int col, int N) {
  const int g = flags >> 16;
  return g ? (int64_t)(row / g) * N + col : col;
}

__device__ __forceinline__ float epilogue(float v, int row, int col, int N, const float* bias, int flags,
                                          float keep, const uint8_t* mask, uint64_t seed) {
  if (flags & F_BIAS) v += bias[bias_index(flags, row, col, N)];
  if (flags & F_RELU) v = fmaxf(v, 0.f);
  if (flags & F_DROP) {
    uint64_t e = (uint64_t)row * (uint64_t)N + (uint64_t)col;
    bool on = mask ? (mask[e] != 0) : (hash_uniform(seed, e) < keep);
    v = on ? v / keep : 0.f;  // tf.nn.dropout: x / keep_prob * binary mask
  }
  if (flags & F_RELUGRAD) {   // backward of relu (+dropout): `mask` carries the layer's fp32 output Y [M,N]
    const float y = reinterpret_cast<const float*>(mask)[(int64_t)row * N + col];
    v = y > 0.f ? v / keep : 0.f;
  }
  return v;
}

// TRANS 0: A[M,K] (lda) , B[K,N] (ldb)
// TRANS 1: A[M,K] (lda) , B[N,K] (ldb)  -> C = A . B^T
// TRANS 2: A[K,M] (lda) , B[K,N] (ldb)  -> C = A^T . B
// Block = 2x2 waves, each wave a (32*WM) x (32*WN) tile => block tile (64*WM) x (64*WN).
template <int TRANS, int WM, int WN, int BK>
__device__ __forceinline__ void gemm_f32_body(const GemmGroup& grp, int blk, const float* __restrict__ bias, int flags,
                                              float keep, const uint8_t* __restrict__ mask, uint64_t seed) {
  // this workgroup's problem and its index inside it (kernels.h: GemmGroup)
  int pi = 0, local = blk;
  while (pi + 1 < grp.n && local >= ((grp.p[pi].nblocks + 7) & ~7)) { local -= (grp.p[pi].nblocks + 7) & ~7; ++pi; }
  const GemmProb& pr = grp.p[pi];
  if (local >= pr.nblocks) return;
  const int M = pr.M, N = pr.N, K = pr.K, lda = pr.lda, ldb = pr.ldb, ldc = pr.ldc, k_chunk = pr.k_chunk;
  const float* __restrict__ A = pr.A;
  const float* __restrict__ Bm = pr.B;
  float* __restrict__ C = pr.C;
  float* __restrict__ slab = pr.slab;
  constexpr int BM = 64 * WM, BN = 64 * WN, BS_LD = BN + 1;
  constexpr int AS_LD = BK + 1;   // As[i][k]: column reads by 32 lanes -> odd word stride, conflict-free
  constexpr int RA = WM * BK / 16, RB = WN * BK / 16, QK = BK / 4;   // staging quads per thread / per k-row
  __shared__ float As[BM * AS_LD];
  __shared__ float Bs[BK * BS_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int bx, by, bz;
  xcd_tile_coords_n(pr.nblocks, local, pr.gx, pr.gy, bx, by, bz);
  const int bm = by * BM, bn = bx * BN;
  const int kbeg = bz * k_chunk;
  const int kend = min(K, kbeg + k_chunk);

  // two staged tiles in flight (register sets 0/1): a grid too small to fill the chip is paced by its K loop, one
  // memory latency per staged tile with a single set
  float ra0[RA][4], rb0[RB][4], ra1[RA][4], rb1[RB][4];
  const bool vecA = ((lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  const bool vecB = ((ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(Bm) & 15) == 0);
  // one row-major quad: 16-B load when aligned and fully in range, guarded scalars otherwise
  auto load_quad = [&](const float* base, int ld, bool vec, int r, int rlim, int c, int clim, float* dst) {
    if (vec && r < rlim && c + 3 < clim) {
      float4 v = ld4(base + (int64_t)r * ld + c);
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) dst[q] = (r < rlim && c + q < clim) ? base[(int64_t)r * ld + c + q] : 0.f;
    }
  };
  auto load_tiles = [&](int k0, float (*ra)[4], float (*rb)[4]) {
#pragma unroll
    for (int rep = 0; rep < RA; ++rep) {
      const int q = tid + 256 * rep;
      if (TRANS == 2) load_quad(A, lda, vecA, k0 + q / (BM / 4), kend, bm + (q % (BM / 4)) * 4, M, ra[rep]);  // A[k][m]
      else            load_quad(A, lda, vecA, bm + q / QK, M, k0 + (q % QK) * 4, kend, ra[rep]);               // A[m][k]
    }
#pragma unroll
    for (int rep = 0; rep < RB; ++rep) {
      const int q = tid + 256 * rep;
      if (TRANS == 1) load_quad(Bm, ldb, vecB, bn + q / QK, N, k0 + (q % QK) * 4, kend, rb[rep]);              // B[n][k]
      else            load_quad(Bm, ldb, vecB, k0 + q / (BN / 4), kend, bn + (q % (BN / 4)) * 4, N, rb[rep]);  // B[k][n]
    }
  };
  auto store_tiles = [&](float (*ra)[4], float (*rb)[4]) {
#pragma unroll
    for (int rep = 0; rep < RA; ++rep) {
      const int q = tid + 256 * rep;
      if (TRANS == 2) {
        const int k = q / (BM / 4), m = (q % (BM / 4)) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) As[(m + e) * AS_LD + k] = ra[rep][e];
      } else {
        const int m = q / QK, k = (q % QK) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) As[m * AS_LD + k + e] = ra[rep][e];
      }
    }
#pragma unroll
    for (int rep = 0; rep < RB; ++rep) {
      const int q = tid + 256 * rep;
      if (TRANS == 1) {
        const int n = q / QK, k = (q % QK) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) Bs[(k + e) * BS_LD + n] = rb[rep][e];
      } else {
        const int k = q / (BN / 4), n = (q % (BN / 4)) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) Bs[k * BS_LD + n + e] = rb[rep][e];
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int arow = wm * 32 * WM + (lane & 31);
  const int bcol = wn * 32 * WN + (lane & 31);
  const int khalf = lane >> 5;
  auto mfma_tile = [&]() {
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float av[WM], bv[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) av[i] = As[(arow + 32 * i) * AS_LD + kk * 2 + khalf];
#pragma unroll
      for (int j = 0; j < WN; ++j) bv[j] = Bs[(kk * 2 + khalf) * BS_LD + bcol + 32 * j];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  };
  if (kbeg < kend) load_tiles(kbeg, ra0, rb0);
  if (kbeg + BK < kend) load_tiles(kbeg + BK, ra1, rb1);
  for (int k0 = kbeg; k0 < kend; k0 += 2 * BK) {
    store_tiles(ra0, rb0);
    __syncthreads();
    if (k0 + 2 * BK < kend) load_tiles(k0 + 2 * BK, ra0, rb0);
    mfma_tile();
    __syncthreads();
    if (k0 + BK >= kend) break;
    store_tiles(ra1, rb1);
    __syncthreads();
    if (k0 + 3 * BK < kend) load_tiles(k0 + 3 * BK, ra1, rb1);
    mfma_tile();
    __syncthreads();
  }

  // C/D layout of a 32x32 accumulator: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = bn + bcol + 32 * j;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = bm + wm * 32 * WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (row >= M) continue;
        if (slab) {
          slab[((int64_t)bz * M + row) * N + col] = acc[i][j][r];
        } else {
          float v = epilogue(acc[i][j][r], row, col, N, pr.bias ? pr.bias : bias, flags, keep, mask, seed);
          float* dst = C + (int64_t)row * ldc + col;
          *dst = (flags & F_ACC) ? *dst + v : v;
        }
      }
    }
}
template <int TRANS, int WM, int WN, int BK>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmGroup grp, const float* __restrict__ bias, int flags,
                                                       float keep, const uint8_t* __restrict__ mask, uint64_t seed) {
  gemm_f32_body<TRANS, WM, WN, BK>(grp, (int)blockIdx.x, bias, flags, keep, mask, seed);
}

__global__ void splitk_reduce_kernel(const float* __restrict__ slab, int nsplit, int M, int N,
                                     float* __restrict__ C, int ldc, const float* __restrict__ bias, int flags,
                                     float keep, const uint8_t* __restrict__ mask, uint64_t seed) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)M * N) return;
  int row = (int)(i / N), col = (int)(i - (int64_t)row * N);
  const int64_t stride = (int64_t)M * N;
  float s = 0.f;
  int z = 0;
  for (; z + 8 <= nsplit; z += 8) {   // 8 independent loads in flight, summed in slab order
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = slab[(int64_t)(z + q) * stride + i];
#pragma unroll
    for (int q = 0; q < 8; ++q) s += v[q];
  }
  for (; z < nsplit; ++z) s += slab[(int64_t)z * stride + i];
  float v = epilogue(s, row, col, N, bias, flags, keep, mask, seed);
  float* dst = C + (int64_t)row * ldc + col;
  *dst = (flags & F_ACC) ? *dst + v : v;
}

// every split-K slab set of a flushed queue in one launch: C = sum_z slab[z]  (slab order)
__device__ __forceinline__ void splitk_reduce_group_body(const ReduceGroup& g, int blk) {
  int ji = 0;
  while (ji + 1 < g.n && blk >= g.j[ji + 1].first_block) ++ji;
  const ReduceJob& job = g.j[ji];
  const int64_t n = (int64_t)job.M * job.N;
  const int64_t i = (int64_t)(blk - job.first_block) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int row = (int)(i / job.N), col = (int)(i - (int64_t)row * job.N);
  float s = 0.f;
  int z = 0;
  for (; z + 8 <= job.ns; z += 8) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = job.slab[(int64_t)(z + q) * n + i];
#pragma unroll
    for (int q = 0; q < 8; ++q) s += v[q];
  }
  for (; z < job.ns; ++z) s += job.slab[(int64_t)z * n + i];
  job.C[(int64_t)row * job.ldc + col] = s;
}
__global__ void splitk_reduce_group_kernel(const ReduceGroup g) { splitk_reduce_group_body(g, (int)blockIdx.x); }

static void fill_prob(GemmProb* p, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                      int ldc, int k_chunk, float* slab, int gx, int gy, int gz) {
  p->A = A; p->B = B; p->C = C; p->slab = slab; p->bias = nullptr; p->M = M; p->N = N; p->K = K; p->lda = lda; p->ldb = ldb;
  p->ldc = ldc; p->k_chunk = k_chunk; p->gx = gx; p->gy = gy; p->nblocks = gx * gy * gz;
}

extern "C" int score_gemm(int32_t trans, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda,
                          const float* Bm, int32_t ldb, float* C, int32_t ldc, const float* bias,
                          int32_t flags, float keep_prob, const uint8_t* drop_mask, uint64_t drop_seed,
                          float* scratch, int64_t scratch_floats, void* stream) {
  if (!A || !Bm || !C || M <= 0 || N <= 0 || K <= 0) return SCORE_E_BADARG;
  if (trans < 0 || trans > 2) return SCORE_E_BADARG;
  if ((flags & F_BIAS) && !bias) return SCORE_E_BADARG;
  if (flags < 0) return SCORE_E_BADARG;
  if ((flags & F_RELUGRAD) && (!drop_mask || (flags & F_DROP))) return SCORE_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  // k-contiguous operands are staged with 16-B loads: need K % 4 == 0, ld % 4 == 0, 16-B aligned base
  const bool a_kc = trans != 2, b_kc = trans == 1;
  // (k-strided operands are staged as 4x4 blocks: their column count must be a multiple of 4 instead)
  const bool al_a = (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
  const bool al_b = (ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(Bm) & 15) == 0;
  const bool x3_ok = al_a && al_b && (a_kc ? (K & 3) == 0 : (M & 3) == 0) && (b_kc ? (K & 3) == 0 : (N & 3) == 0);
  // measured on MI355X (tools/gemm_ab.py, interleaved A/B in one process): the split pays where the
  // tile is wide and the K loop long; small products stay on the f32 MFMA kernel
  const bool x3_shape = (trans == 0 && M >= 4096 && N >= 64) ||
                        (trans == 1 && M >= 4096 && N >= 256) ||
                        (trans == 2 && (int64_t)M * N >= 32768 && K >= 4096);
  const bool x3_force = (flags & 32) != 0;     // tests: take the bf16x3 kernel whenever it is legal
  if ((flags & (F_X3 | 32)) && x3_ok && M >= 64 && N >= 32 && K >= 32 && (x3_shape || x3_force)) {
    // fp32-accurate product on the bf16 matrix cores (gemm_bf16x3.hip): (64*wm) x 128 x 32 tiles.
    // Pick the tile height by how many tiles the busiest CU gets (co-resident blocks share its matrix
    // pipe, so what counts is tiles per CU, not per residency slot); a 64-row tile costs ~57 % of a
    // 128-row one (same B tile, half the MFMAs).
    auto rounds = [&](int wm_) {
      int64_t blocks = (int64_t)((N + 127) / 128) * ((M + 64 * wm_ - 1) / (64 * wm_));
      return (double)((blocks + 255) / 256) * (wm_ == 2 ? 1.0 : 0.57);
    };
    const int wm3 = rounds(1) < rounds(2) ? 1 : 2;
    dim3 g3((N + 127) / 128, (M + 64 * wm3 - 1) / (64 * wm3), 1);
    int ns = 1;
    int64_t t3 = (int64_t)g3.x * g3.y;
    if (t3 < 256 && K >= 512 && scratch) {
      ns = (int)((320 + t3 - 1) / t3);
      if (ns > K / 128) ns = K / 128;
      while (ns > 1 && (int64_t)ns * M * N > scratch_floats) --ns;
      if (ns < 1) ns = 1;
    }
    int kc = K;
    float* sl = nullptr;
    if (ns > 1) {
      kc = (int)align_up64(cdiv64(K, ns), 32);
      ns = (int)cdiv64(K, kc);
      g3.z = ns;
      sl = ns > 1 ? scratch : nullptr;
    }
    GemmGroup grp;
    grp.n = 1;
    fill_prob(&grp.p[0], M, N, K, A, lda, Bm, ldb, C, ldc, kc, sl, (int)g3.x, (int)g3.y, (int)g3.z);
    grp.total_blocks = (grp.p[0].nblocks + 7) & ~7;
    SCORE_TRY(score_launch_gemm_bf16x3(trans, wm3, grp, bias, flags & KF_MASK, keep_prob, drop_mask, drop_seed, s));
    if (sl) {
      int64_t n = (int64_t)M * N;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, sl, ns, M, N, C, ldc,
                         bias, flags & KF_MASK, keep_prob, drop_mask, drop_seed);
      SCORE_CHECK_LAUNCH();
    }
    return 0;
  }
  flags &= KF_MASK;
  // tile choice: the largest wave tile that still gives the chip >= ~1.5 blocks per CU
  int WMs = 1, WNs = 1;
  auto nblocks = [&](int wm_, int wn_) { return (int64_t)((N + 64 * wn_ - 1) / (64 * wn_)) * ((M + 64 * wm_ - 1) / (64 * wm_)); };
  const int64_t want = 384;
  const int64_t ksplit_cap = (K >= GEMM_SPLITK_MIN_K && scratch) ? K / GEMM_SPLITK_MIN_CHUNK : 1;
  if (N > 64 && nblocks(1, 2) * ksplit_cap >= want) WNs = 2;
  if (M > 64 && nblocks(2, WNs) * ksplit_cap >= want) WMs = 2;
  const int BMh = 64 * WMs, BNh = 64 * WNs;
  dim3 grid((N + BNh - 1) / BNh, (M + BMh - 1) / BMh, 1);
  // split the reduced dimension when the output grid alone cannot fill 256 CUs
  int nsplit = 1;
  int64_t tiles = (int64_t)grid.x * grid.y;
  if (tiles < 256 && K >= GEMM_SPLITK_MIN_K && scratch) {
    nsplit = (int)((320 + tiles - 1) / tiles);   // ~1.25 blocks per CU: longer K chunks, smaller slabs
    int max_split = K / GEMM_SPLITK_MIN_CHUNK;
    if (nsplit > max_split) nsplit = max_split;
    while (nsplit > 1 && (int64_t)nsplit * M * N > scratch_floats) --nsplit;
    if (nsplit < 1) nsplit = 1;
  }
  int k_chunk = K;
  float* slab = nullptr;
  if (nsplit > 1) {
    k_chunk = (int)align_up64(cdiv64(K, nsplit), BK_ALIGN);
    nsplit = (int)cdiv64(K, k_chunk);
    grid.z = nsplit;
    slab = nsplit > 1 ? scratch : nullptr;
  }
  GemmGroup grp;
  grp.n = 1;
  fill_prob(&grp.p[0], M, N, K, A, lda, Bm, ldb, C, ldc, k_chunk, slab, (int)grid.x, (int)grid.y, (int)grid.z);
  grp.total_blocks = (grp.p[0].nblocks + 7) & ~7;
#define LAUNCH(TR, WMv, WNv)                                                                                       \
  hipLaunchKernelGGL((gemm_f32_kernel<TR, WMv, WNv, 16>), dim3(grp.total_blocks), dim3(256), 0, s, grp, bias, flags, \
                     keep_prob, drop_mask, drop_seed)
#define LAUNCH_T(TR)                                      \
  do {                                                    \
    if (WMs == 2 && WNs == 2) LAUNCH(TR, 2, 2);           \
    else if (WMs == 2) LAUNCH(TR, 2, 1);                  \
    else if (WNs == 2) LAUNCH(TR, 1, 2);                  \
    else LAUNCH(TR, 1, 1);                                \
  } while (0)
  if (trans == 0) LAUNCH_T(0);
  else if (trans == 1) LAUNCH_T(1);
  else LAUNCH_T(2);
#undef LAUNCH_T
#undef LAUNCH
  SCORE_CHECK_LAUNCH();
  if (slab) {
    int64_t n = (int64_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, slab, nsplit, M, N,
                       C, ldc, bias, flags, keep_prob, drop_mask, drop_seed);
    SCORE_CHECK_LAUNCH();
  }
  return 0;
}

// nprob same-shape problems in one grouped launch (no bias, no split-K: meant for per-time-slice products whose
// tiles already fill the chip together); anything it cannot take goes through score_gemm one by one.
int score_gemm_same_shape(int trans, int nprob, int M, int N, int K, const float* const* A, int lda,
                          const float* const* B, int ldb, float* const* C, int ldc, int flags, int x3, float* scratch,
                          int64_t scratch_floats, hipStream_t s, const float* const* bias) {
  if (nprob <= 0 || nprob > GEMM_GROUP_MAX || trans < 0 || trans > 2 || (flags & ~(F_ACC | F_BIAS))) return SCORE_E_BADARG;
  if ((flags & F_BIAS) && !bias) return SCORE_E_BADARG;
  bool al = (lda & 3) == 0 && (ldb & 3) == 0;
  for (int i = 0; i < nprob; ++i)
    al = al && (reinterpret_cast<uintptr_t>(A[i]) & 15) == 0 && (reinterpret_cast<uintptr_t>(B[i]) & 15) == 0;
  const bool a_kc = trans != 2, b_kc = trans == 1;
  const bool x3_ok = al && (a_kc ? (K & 3) == 0 : (M & 3) == 0) && (b_kc ? (K & 3) == 0 : (N & 3) == 0) && M >= 64 &&
                     N >= 32 && K >= 32;
  const bool x3_shape = (trans == 0 && M >= 4096 && N >= 64) || (trans == 1 && M >= 4096 && N >= 256);
  GemmGroup grp;
  grp.n = nprob; grp.total_blocks = 0;
  if (x3 && x3_ok && x3_shape) {
    const int64_t t2 = (int64_t)nprob * ((N + 127) / 128) * ((M + 127) / 128);
    const int64_t t1 = (int64_t)nprob * ((N + 127) / 128) * ((M + 63) / 64);
    auto rounds = [](int64_t blocks, double w) { return (double)((blocks + 255) / 256) * w; };
    const int wm = rounds(t1, 0.57) < rounds(t2, 1.0) ? 1 : 2;
    if ((wm == 2 ? t2 : t1) >= 128) {
      for (int i = 0; i < nprob; ++i) {
        fill_prob(&grp.p[i], M, N, K, A[i], lda, B[i], ldb, C[i], ldc, K, nullptr, (N + 127) / 128,
                  (M + 64 * wm - 1) / (64 * wm), 1);
        grp.p[i].bias = bias ? bias[i] : nullptr;
        grp.total_blocks += (grp.p[i].nblocks + 7) & ~7;
      }
      return score_launch_gemm_bf16x3(trans, wm, grp, nullptr, flags, 1.f, nullptr, 0, s);
    }
  }
  const int64_t tiles = (int64_t)nprob * ((N + 63) / 64) * ((M + 63) / 64);
  if (tiles < 128 && K >= GEMM_SPLITK_MIN_K && scratch) {          // too few tiles even together: let score_gemm split K
    // (with a K too short to split the separate launches gain nothing: the small shapes' projections -- 2 x 58 tiles, K = 112
    //  at the reference's own shape -- go out together below)
    for (int i = 0; i < nprob; ++i)
      SCORE_TRY(score_gemm(trans, M, N, K, A[i], lda, B[i], ldb, C[i], ldc, bias ? bias[i] : nullptr, flags | (x3 ? F_X3 : 0),
                           1.f, nullptr, 0, scratch, scratch_floats, s));
    return 0;
  }
  for (int i = 0; i < nprob; ++i) {
    fill_prob(&grp.p[i], M, N, K, A[i], lda, B[i], ldb, C[i], ldc, K, nullptr, (N + 63) / 64, (M + 63) / 64, 1);
    grp.p[i].bias = bias ? bias[i] : nullptr;
    grp.total_blocks += (grp.p[i].nblocks + 7) & ~7;
  }
#define LS(TR) hipLaunchKernelGGL((gemm_f32_kernel<TR, 1, 1, 16>), dim3(grp.total_blocks), dim3(256), 0, s, grp, nullptr, \
                                  flags, 1.f, nullptr, 0)
  if (trans == 0) LS(0); else if (trans == 1) LS(1); else LS(2);
#undef LS
  SCORE_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ deferred weight-gradient products
int gemm_queue_add(GemmQueue* q, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                   int ldc) {
  if (!q || !A || !B || !C || M <= 0 || N <= 0 || K <= 0) return SCORE_E_BADARG;
  if (q->n >= 2 * GEMM_GROUP_MAX) return SCORE_E_WORKSPACE;
  GemmQueueJob& j = q->j[q->n++];
  j.A = A; j.B = B; j.C = C; j.M = M; j.N = N; j.K = K; j.lda = lda; j.ldb = ldb; j.ldc = ldc;
  return 0;
}

// C_j = A_j^T . B_j for every queued job: the jobs that qualify for the bf16x3 kernel go out as one grouped
// launch of 128x128 tiles, the rest as one grouped launch of the 64x64 f32 kernel, K split so that either
// launch fills the chip a few times over; one more launch reduces all slabs (fixed order: reproducible).
__global__ void gemm_colsum_kernel(const GemmGroup grp, const ColsumJobs jobs, float* __restrict__ part, int gemm_blocks, int gx);
int gemm_queue_flush(GemmQueue* q, int x3, float* slab, int64_t slab_floats, hipStream_t s, ReduceGroup* defer,
                     const ColsumJobs* with_colsums, float* cs_part, int64_t cs_part_floats, int* colsums_done) {
  if (colsums_done) *colsums_done = 0;
  if (defer) defer->n = defer->blocks = 0;
  if (!q || q->n == 0) return 0;
  if (!slab) return SCORE_E_BADARG;
  GemmGroup g3, gf;
  g3.n = gf.n = 0; g3.total_blocks = gf.total_blocks = 0;
  int fam[2 * GEMM_GROUP_MAX];
  int64_t work[2] = {0, 0};          // sum over jobs of tiles * K, per family (0 = f32, 1 = bf16x3)
  for (int i = 0; i < q->n; ++i) {
    const GemmQueueJob& j = q->j[i];
    const bool al = (j.lda & 3) == 0 && (j.ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(j.A) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(j.B) & 15) == 0 && (j.M & 3) == 0 && (j.N & 3) == 0;
    fam[i] = (x3 && al && j.M >= 64 && j.N >= 32 && (int64_t)j.M * j.N >= 16384 && j.K >= 4096) ? 1 : 0;
    const int64_t tiles = fam[i] ? (int64_t)((j.M + 127) / 128) * ((j.N + 127) / 128)
                                 : (int64_t)((j.M + 63) / 64) * ((j.N + 63) / 64);
    work[fam[i]] += tiles * j.K;
  }
  // K chunk per family: ~512 (bf16x3, two 128x128 blocks per CU: both operands stream, so the larger tile
  // halves the bytes pulled through L2 per flop) / ~1024 (f32) blocks in flight
  int kc[2];
  kc[1] = (int)align_up64(cdiv64(work[1] > 0 ? work[1] : 1, 512), 32);
  if (kc[1] < 256) kc[1] = 256;
  // ... and not ONE workgroup more than the 512 that are resident together (every job's share is padded to a multiple of
  // eight below): a 513th starts a second round of the whole launch -- measured +27 us on a 100-us launch at cfg-3
  for (int guard = 0; guard < 256; ++guard) {
    int64_t blocks = 0;
    for (int i = 0; i < q->n; ++i)
      if (fam[i]) {
        const GemmQueueJob& j = q->j[i];
        blocks += ((int64_t)((j.M + 127) / 128) * ((j.N + 127) / 128) * cdiv64(j.K, kc[1]) + 7) & ~(int64_t)7;
      }
    if (blocks <= 512 || kc[1] >= 1 << 24) break;
    kc[1] += (int)align_up64(kc[1] / 64 > 32 ? kc[1] / 64 : 32, 32);       // (K = 204,800 at cfg-5: steps of 32 would never get there)
  }
  kc[0] = (int)align_up64(cdiv64(work[0] > 0 ? work[0] : 1, 1024), BK_ALIGN);
  if (kc[0] < 128) kc[0] = 128;
  ReduceGroup rg;
  rg.n = 0;
  int rblocks = 0;
  int64_t used = 0;
  for (int i = 0; i < q->n; ++i) {
    const GemmQueueJob& j = q->j[i];
    GemmGroup& g = fam[i] ? g3 : gf;
    if (g.n >= GEMM_GROUP_MAX) return SCORE_E_WORKSPACE;
    const int gx = fam[i] ? (j.N + 127) / 128 : (j.N + 63) / 64, gy = fam[i] ? (j.M + 127) / 128 : (j.M + 63) / 64;
    int chunk = kc[fam[i]];
    int ns = (int)cdiv64(j.K, chunk);
    const int64_t mn = (int64_t)j.M * j.N;
    while (ns > 1 && used + (int64_t)ns * mn > slab_floats) {   // not enough slab room: longer chunks
      chunk *= 2;
      ns = (int)cdiv64(j.K, chunk);
    }
    float* sl = nullptr;
    if (ns > 1) {
      sl = slab + used;
      used += align_up64((int64_t)ns * mn, 4);
      ReduceJob& r = rg.j[rg.n++];
      r.slab = sl; r.C = j.C; r.ns = ns; r.M = j.M; r.N = j.N; r.ldc = j.ldc; r.first_block = rblocks; r.pad = 0;
      rblocks += (int)cdiv64(mn, 256);
    } else {
      chunk = j.K;
    }
    fill_prob(&g.p[g.n], j.M, j.N, j.K, j.A, j.lda, j.B, j.ldb, j.C, j.ldc, chunk, sl, gx, gy, ns);
    g.total_blocks += (g.p[g.n].nblocks + 7) & ~7;
    ++g.n;
  }
  if (g3.n) SCORE_TRY(score_launch_gemm_bf16x3(2, 2, g3, nullptr, 0, 1.f, nullptr, 0, s));
  if (gf.n && with_colsums && with_colsums->n > 0 && cs_part && with_colsums->part_used <= cs_part_floats) {
    int gx = 1;
    for (int i = 0; i < with_colsums->n; ++i)
      gx = max(gx, (with_colsums->job[i].N + with_colsums->job[i].cols - 1) / with_colsums->job[i].cols);
    const int64_t cs_blocks = (int64_t)gx * COLSUM_MAX_PARTS * with_colsums->n;
    hipLaunchKernelGGL(gemm_colsum_kernel, dim3((unsigned)(gf.total_blocks + cs_blocks)), dim3(256), 0, s, gf, *with_colsums, cs_part,
                       gf.total_blocks, gx);
    SCORE_CHECK_LAUNCH();
    if (colsums_done) *colsums_done = 1;
  } else if (gf.n) {
    hipLaunchKernelGGL((gemm_f32_kernel<2, 1, 1, 16>), dim3(gf.total_blocks), dim3(256), 0, s, gf, nullptr, 0, 1.f, nullptr,
                       0);
    SCORE_CHECK_LAUNCH();
  }
  rg.blocks = rblocks;
  if (defer) {
    *defer = rg;                // (the caller launches it: score_launch_finish)
  } else if (rg.n) {
    hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(rblocks), dim3(256), 0, s, rg);
    SCORE_CHECK_LAUNCH();
  }
  q->n = 0;
  return 0;
}

// ------------------------------------------------------------------ small helpers used by the engine
// out[n] (+)= sum_m X[m][n]   two-stage, fixed order.  A block is (cols x rows) threads;
// each thread strides down its column, the row dimension is combined through LDS.
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ X, int M, int N, int ld, int cols,
                                                     int rows_per_block, float* __restrict__ part) {
  __shared__ float sh[256];
  const int tx = threadIdx.x % cols, ty = threadIdx.x / cols, nty = 256 / cols;
  const int n = blockIdx.x * cols + tx;
  const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float s = 0.f;
  if (n < N)
    for (int m = m0 + ty; m < m1; m += nty) s += X[(int64_t)m * ld + n];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (ty == 0 && n < N) {
    float t = 0.f;
    for (int r = 0; r < nty; ++r) t += sh[r * cols + tx];
    part[(int64_t)blockIdx.y * N + n] = t;
  }
}
__global__ void colsum_stage2(const float* __restrict__ part, int nparts, int N, float* __restrict__ out,
                              int accumulate) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += part[(int64_t)p * N + n];
  out[n] = accumulate ? out[n] + s : s;
}

int score_launch_colsum(const float* X, int M, int N, int ld, float* out, int accumulate, float* scratch,
                        int64_t scratch_floats, hipStream_t s) {
  int cols = 1;
  while (cols < N && cols < 64) cols <<= 1;
  // <= 32 partial rows so the fixed-order second stage stays a short dependent chain
  int rpb = 128;
  int nparts = (M + rpb - 1) / rpb;
  if (nparts > 32) { nparts = 32; rpb = (M + nparts - 1) / nparts; nparts = (M + rpb - 1) / rpb; }
  if ((int64_t)nparts * N > scratch_floats) return SCORE_E_WORKSPACE;
  hipLaunchKernelGGL(colsum_stage1, dim3((N + cols - 1) / cols, nparts), dim3(256), 0, s, X, M, N, ld, cols, rpb,
                     scratch);
  SCORE_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_stage2, dim3((N + 63) / 64), dim3(64), 0, s, scratch, nparts, N, out, accumulate);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// ---- many column sums in two launches (bias gradients and slab reductions of one backward pass)
__device__ __forceinline__ void colsum_multi_stage1_body(const ColsumJobs& jobs, float* __restrict__ part, int bx, int by, int bz) {
  __shared__ float sh[256];
  const ColsumJob& j = jobs.job[bz];
  const int cols = j.cols, nty = 256 / cols;
  if (by >= j.nparts || bx * cols >= j.N) return;
  const int tx = threadIdx.x % cols, ty = threadIdx.x / cols;
  const int n = bx * cols + tx;
  const int m0 = by * j.rpb, m1 = min(j.M, m0 + j.rpb);
  float s = 0.f;
  if (n < j.N) {
    // four rows in flight per thread; the order of the adds is fixed by (rpb, nty): reproducible
    const float* x = j.X + n;
    int m = m0 + ty;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j.scale) {          // rows weighted by scale[m] (X^T s: the co-attention's target-row weight gradients)
      const float* sc = j.scale;
      for (; m + 3 * nty < m1; m += 4 * nty) {
        const float v0 = x[(int64_t)m * j.ld], v1 = x[(int64_t)(m + nty) * j.ld];
        const float v2 = x[(int64_t)(m + 2 * nty) * j.ld], v3 = x[(int64_t)(m + 3 * nty) * j.ld];
        const float c0 = sc[m], c1 = sc[m + nty], c2 = sc[m + 2 * nty], c3 = sc[m + 3 * nty];
        s0 = fmaf(v0, c0, s0); s1 = fmaf(v1, c1, s1); s2 = fmaf(v2, c2, s2); s3 = fmaf(v3, c3, s3);
      }
      for (; m < m1; m += nty) s0 = fmaf(x[(int64_t)m * j.ld], sc[m], s0);
    } else {
      for (; m + 3 * nty < m1; m += 4 * nty) {
        const float v0 = x[(int64_t)m * j.ld], v1 = x[(int64_t)(m + nty) * j.ld];
        const float v2 = x[(int64_t)(m + 2 * nty) * j.ld], v3 = x[(int64_t)(m + 3 * nty) * j.ld];
        s0 += v0; s1 += v1; s2 += v2; s3 += v3;
      }
      for (; m < m1; m += nty) s0 += x[(int64_t)m * j.ld];
    }
    s = (s0 + s1) + (s2 + s3);
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (ty == 0 && n < j.N) {
    float t = 0.f;
    for (int r = 0; r < nty; ++r) t += sh[r * cols + tx];
    part[j.part_off + (int64_t)by * j.N + n] = t;
  }
}
__global__ __launch_bounds__(256) void colsum_multi_stage1(const ColsumJobs jobs, float* __restrict__ part) {
  colsum_multi_stage1_body(jobs, part, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}
// The grouped f32 weight-gradient products AND the column sums' first stage in ONE launch (round 5, the per-sample form): both
// read what the backward kernel left, neither reads the other -- as two launches on one stream they ran one after the other
// (16 + 17 us at the Tmall default shape, in front of the dense variables' ApplyAdam).  Workgroups [0, gemm_blocks): the
// products (gemm_f32_kernel<2, 1, 1, 16>'s body), the rest: colsum_multi_stage1's (bx fastest, then part, then job).
__global__ __launch_bounds__(256) void gemm_colsum_kernel(const GemmGroup grp, const ColsumJobs jobs, float* __restrict__ part,
                                                          int gemm_blocks, int gx) {
  const int blk = (int)blockIdx.x;
  if (blk < gemm_blocks) {
    gemm_f32_body<2, 1, 1, 16>(grp, blk, nullptr, 0, 1.f, nullptr, 0);
    return;
  }
  const int b = blk - gemm_blocks;
  colsum_multi_stage1_body(jobs, part, b % gx, (b / gx) % COLSUM_MAX_PARTS, b / (gx * COLSUM_MAX_PARTS));
}
__device__ __forceinline__ void colsum_multi_stage2_body(const ColsumJobs& jobs, const float* __restrict__ part, int job, int bx) {
  const ColsumJob& j = jobs.job[job];
  int n = bx * blockDim.x + threadIdx.x;
  if (n >= j.N) return;
  const float* p0 = part + j.part_off + n;
  float s = 0.f;
  int p = 0;
  for (; p + 8 <= j.nparts; p += 8) {   // 8 loads in flight, summed in part order
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = p0[(int64_t)(p + q) * j.N];
#pragma unroll
    for (int q = 0; q < 8; ++q) s += v[q];
  }
  for (; p < j.nparts; ++p) s += p0[(int64_t)p * j.N];
  j.out[n] = j.acc ? j.out[n] + s : s;
}
__global__ void colsum_multi_stage2(const ColsumJobs jobs, const float* __restrict__ part) {
  colsum_multi_stage2_body(jobs, part, (int)blockIdx.y, (int)blockIdx.x);
}
// workgroups [0, rg.blocks): the split-K slab reduce; the rest, gx2 per job: the column sums' second stage
// one element of a product from its split-K slabs, in slab order (splitk_reduce_group_body's sum), or from C where K was not split
__device__ __forceinline__ float w1_piece(const float* __restrict__ slab, int ns, int64_t n, const float* __restrict__ C, int64_t i) {
  if (ns <= 1) return C[i];
  float s = 0.f;
  for (int z = 0; z < ns; ++z) s += slab[(int64_t)z * n + i];
  return s;
}
__global__ __launch_bounds__(256) void finish_kernel(const ReduceGroup rg, const ColsumJobs jobs, const float* __restrict__ part,
                                                     int gx2, int cs_blocks, const W1Fold w1) {
  const int blk = (int)blockIdx.x;
  if (blk < rg.blocks) {
    splitk_reduce_group_body(rg, blk);
    return;
  }
  const int b = blk - rg.blocks;
  if (b < cs_blocks) {
    colsum_multi_stage2_body(jobs, part, b / gx2, b % gx2);
    return;
  }
  // the folded first attention layer's gradient (head.hip attn_w1_grad_kernel: dWa = dWq, dWb = dWeff_k, dWc = dWq - dWeff_k,
  // dWd = dWeff_qk) straight from the two products' slabs -- it was a launch of its own behind this one
  const int64_t n = (int64_t)w1.Dk * w1.NA;
  const int64_t i = (int64_t)(b - cs_blocks) * 256 + threadIdx.x;
  if (i >= n) return;
  const float dq = w1_piece(w1.slab_q, w1.ns_q, n, w1.dwq, i);
  const float dk = w1_piece(w1.slab_e, w1.ns_e, 2 * n, w1.dweff, i);
  const float dqk = w1_piece(w1.slab_e, w1.ns_e, 2 * n, w1.dweff, n + i);
  w1.gW1[i] = dq;
  w1.gW1[n + i] = dk;
  w1.gW1[2 * n + i] = dq - dk;
  w1.gW1[3 * n + i] = dqk;
}

int colsum_queue_add(ColsumJobs* q, const float* X, int M, int N, int ld, float* out, int acc, const float* scale) {
  if (q->n >= COLSUM_MAX_JOBS) return SCORE_E_WORKSPACE;
  ColsumJob& j = q->job[q->n++];
  j.X = X; j.M = M; j.N = N; j.ld = ld; j.out = out; j.acc = acc; j.scale = scale;
  int cols = 1;
  while (cols < N && cols < 64) cols <<= 1;
  j.cols = cols;
  // up to COLSUM_MAX_PARTS row slices per job: a [B*T, 3H] bias gradient alone then fills the chip
  int rpb = 64, nparts = (M + rpb - 1) / rpb;
  if (nparts > COLSUM_MAX_PARTS) {
    nparts = COLSUM_MAX_PARTS; rpb = (M + nparts - 1) / nparts; nparts = (M + rpb - 1) / rpb;
  }
  j.rpb = rpb; j.nparts = nparts;
  j.part_off = q->part_used;
  q->part_used += (int64_t)nparts * N;
  return 0;
}

int score_launch_finish(const ReduceGroup* rg, ColsumJobs* q, float* part, int64_t part_floats, hipStream_t s, int stage1_done,
                        const W1Fold* w1) {
  ReduceGroup none;
  none.n = none.blocks = 0;
  if (!rg) rg = &none;
  W1Fold wf;
  memset(&wf, 0, sizeof(wf));
  int w1_blocks = 0;
  if (w1 && q && q->n > 0) {
    wf = *w1;
    for (int i = 0; i < rg->n; ++i) {       // the two products' slab sets (none: K was not split, the product wrote its C)
      if (rg->j[i].C == w1->dweff) { wf.slab_e = rg->j[i].slab; wf.ns_e = rg->j[i].ns; }
      if (rg->j[i].C == w1->dwq) { wf.slab_q = rg->j[i].slab; wf.ns_q = rg->j[i].ns; }
    }
    w1_blocks = (int)cdiv64((int64_t)w1->Dk * w1->NA, 256);
  } else if (w1) {
    return SCORE_E_BADARG;
  }
  if (!q || q->n == 0) {
    if (rg->n) {
      hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(rg->blocks), dim3(256), 0, s, *rg);
      SCORE_CHECK_LAUNCH();
    }
    return 0;
  }
  if (q->part_used > part_floats) return SCORE_E_WORKSPACE;
  int gx = 1, gx2 = 1;
  for (int i = 0; i < q->n; ++i) {
    gx = max(gx, (q->job[i].N + q->job[i].cols - 1) / q->job[i].cols);
    gx2 = max(gx2, (q->job[i].N + 255) / 256);
  }
  if (!stage1_done) {
    hipLaunchKernelGGL(colsum_multi_stage1, dim3(gx, COLSUM_MAX_PARTS, q->n), dim3(256), 0, s, *q, part);
    SCORE_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(finish_kernel, dim3(rg->blocks + gx2 * q->n + w1_blocks), dim3(256), 0, s, *rg, *q, part, gx2, gx2 * q->n, wf);
  SCORE_CHECK_LAUNCH();
  q->n = 0;
  q->part_used = 0;
  return 0;
}

// the first stage alone (the queue stays: score_launch_finish(..., stage1_done = 1) runs the second stage and empties it)
int colsum_queue_stage1(const ColsumJobs* q, float* part, int64_t part_floats, hipStream_t s) {
  if (!q || q->n == 0) return 0;
  if (q->part_used > part_floats) return SCORE_E_WORKSPACE;
  int gx = 1;
  for (int i = 0; i < q->n; ++i) gx = max(gx, (q->job[i].N + q->job[i].cols - 1) / q->job[i].cols);
  hipLaunchKernelGGL(colsum_multi_stage1, dim3(gx, COLSUM_MAX_PARTS, q->n), dim3(256), 0, s, *q, part);
  SCORE_CHECK_LAUNCH();
  return 0;
}

int colsum_queue_flush(ColsumJobs* q, float* part, int64_t part_floats, hipStream_t s) {
  if (q->n == 0) return 0;
  if (q->part_used > part_floats) return SCORE_E_WORKSPACE;
  int gx = 1, gx2 = 1;
  for (int i = 0; i < q->n; ++i) {
    gx = max(gx, (q->job[i].N + q->job[i].cols - 1) / q->job[i].cols);
    gx2 = max(gx2, (q->job[i].N + 63) / 64);
  }
  hipLaunchKernelGGL(colsum_multi_stage1, dim3(gx, COLSUM_MAX_PARTS, q->n), dim3(256), 0, s, *q, part);
  SCORE_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_multi_stage2, dim3(gx2, q->n), dim3(64), 0, s, *q, part);
  SCORE_CHECK_LAUNCH();
  q->n = 0;
  q->part_used = 0;
  return 0;
}
