// fp32-accurate GEMM on the bf16 matrix cores ("bf16x3").
//
// gfx950's f32-input MFMA runs at the VALU rate (64 FLOP/clk/SIMD = 1/16 of bf16) and there is no
// xf32.  An fp32 value splits EXACTLY into three bf16 pieces (8+8+8 significand bits, by truncation):
// x = x0 + x1 + x2.  x*y = sum_{p,q} x_p y_q; every bf16 product is exact in fp32, and the three
// dropped terms (x1 y2, x2 y1, x2 y2) are below 2^-24 |x y| -- one fp32 rounding.  So six
// v_mfma_f32_32x32x16_bf16 per k-step reproduce an fp32 GEMM to fp32 accuracy at 16/6 = 2.7x the
// f32-MFMA rate.  The split is done once per element while staging the tile into LDS.
//
// Tile: (64*WM)x128x32 per 256-thread block, 2x2 waves of (32*WM)x64, two blocks per CU.  LDS holds the
// three bf16 planes of A as [m][k] and of B as [n][k] (k contiguous, 16-B fragments, row stride 40
// bf16 = 80 B: conflict-free ds_read_b128).  The K loop is software-pipelined by hand: each of the
// tile's MFMA slots carries its share of the NEXT tile's split (a few VALU ops, in the shadow of the
// 32-cycle MFMA), of the tile-after-next's global loads (issued as soon as the split frees their
// registers) and of the second k-step's fragment reads; sched_barrier pins the interleave.  Workgroups
// take their tile through the XCD-aware order of common.h.  Same operand layouts / epilogue / split-K
// as gemm.hip.  X3_PROBE_* macros strip one ingredient at a time for tools/x3_probe.py.
#include <type_traits>
#include "common.h"
#include "kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TBN 128
#define TBK 32
#define TLD 40   // bf16 per LDS row (32 + 8 pad)

enum { XF_BIAS = 1, XF_RELU = 2, XF_ACC = 4, XF_DROP = 8, XF_RELUGRAD = 64 };

__device__ __forceinline__ float x3_epilogue(float v, int row, int col, int N, const float* bias, int flags,
                                             float keep, const uint8_t* mask, uint64_t seed) {
  if (flags & XF_BIAS) {
    const int g = flags >> 16;   // bias row group (score_gemm)
    v += bias[g ? (int64_t)(row / g) * N + col : col];
  }
  if (flags & XF_RELU) v = fmaxf(v, 0.f);
  if (flags & XF_DROP) {
    uint64_t e = (uint64_t)row * (uint64_t)N + (uint64_t)col;
    bool on = mask ? (mask[e] != 0) : (hash_uniform(seed, e) < keep);
    v = on ? v / keep : 0.f;
  }
  if (flags & XF_RELUGRAD) {   // backward of relu (+dropout): `mask` carries the layer's fp32 output Y [M,N]
    const float y = reinterpret_cast<const float*>(mask)[(int64_t)row * N + col];
    v = y > 0.f ? v / keep : 0.f;
  }
  return v;
}

// exact 3-way split: returns the three bf16 bit patterns (upper halves of fp32 words)
__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  uint32_t xb = __float_as_uint(x);
  h = xb & 0xFFFF0000u;
  float r1 = x - __uint_as_float(h);
  uint32_t rb = __float_as_uint(r1);
  m = rb & 0xFFFF0000u;
  float r2 = r1 - __uint_as_float(m);
  l = __float_as_uint(r2);       // (pack2 keeps the upper half only)
}
// pack the upper halves of two fp32 words (= two bf16) into one dword, first element low: one v_perm_b32
__device__ __forceinline__ uint32_t pack2(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// WM = 32-row MFMA tiles per wave along M: block tile (64*WM) x 128
template <int TRANS, int WM>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(const GemmGroup grp, const float* __restrict__ bias, int flags,
                                                             float keep, const uint8_t* __restrict__ mask,
                                                             uint64_t seed) {
  // this workgroup's problem and its index inside it
  int pi = 0, local = (int)blockIdx.x;
  while (pi + 1 < grp.n && local >= ((grp.p[pi].nblocks + 7) & ~7)) { local -= (grp.p[pi].nblocks + 7) & ~7; ++pi; }
  const GemmProb& pr = grp.p[pi];
  if (local >= pr.nblocks) return;
  const int M = pr.M, N = pr.N, K = pr.K, lda = pr.lda, ldb = pr.ldb, ldc = pr.ldc, k_chunk = pr.k_chunk;
  const float* __restrict__ A = pr.A;
  const float* __restrict__ Bm = pr.B;
  float* __restrict__ C = pr.C;
  float* __restrict__ slab = pr.slab;
  constexpr int TBM = 64 * WM;
  constexpr int EA = TBM * TBK / 256;      // A elements staged per thread (16 or 8)
  __shared__ __attribute__((aligned(16))) unsigned short Ap[3][TBM * TLD];
  __shared__ __attribute__((aligned(16))) unsigned short Bp[3][TBN * TLD];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int bx, by, bz;
  xcd_tile_coords_n(pr.nblocks, local, pr.gx, pr.gy, bx, by, bz);
  const int bm = by * TBM, bn = bx * TBN;
  const int kbeg = bz * k_chunk;
  const int kend = min(K, kbeg + k_chunk);
  constexpr bool A_KCONTIG = (TRANS != 2);   // A[m][k]
  constexpr bool B_KCONTIG = (TRANS == 1);   // B[n][k]

  // Register pipeline: tile t is in LDS, tile t+1 is being split (VALU, interleaved with the MFMAs of
  // tile t), tile t+2 goes in flight from global memory as its staging registers are consumed.
  // Per-thread element offsets are fixed for the whole K loop (computed once; rows/columns past the
  // edge are clamped onto the last valid one -- their products land in C rows/columns that are never
  // stored); only the K tail needs zero fill, and only the last tile(s) take that path.
  constexpr int XA = A_KCONTIG ? EA : 16;  // a k-strided operand is staged as 4x4 blocks: 16 per active thread
  constexpr int NA4 = XA / 4;              // 16-B loads per thread and tile
  float xa[XA], xb[16];                    // staging registers (tile t+1, then t+2)
  uint32_t pa[3 * XA / 2], pb[24];
  uint32_t offa[NA4], offb[4];             // element offsets of this thread's quads inside a k-tile
  int ka[NA4], kb4[4];                     // k of each quad relative to the tile start
  // k-contiguous operand X[r][k]: quad q = tid + 256*rep: row q>>3, k (q&7)*4
  // k-strided operand X[k][c]: a 4(k) x 4(c) block: k-group tid&7, column-group tid>>3 (rep = k inside the block)
  const bool a_mine = A_KCONTIG || (tid >> 3) * 4 < TBM;   // a 64-wide k-strided tile has only 128 blocks of 4x4
#pragma unroll
  for (int rep = 0; rep < NA4; ++rep) {
    if (A_KCONTIG) {
      // two adjacent quads per thread (8 consecutive k of one row): their three planes go to LDS as one 16-B
      // write each instead of two 8-B ones (LDS writes cost 11.5 of this kernel's 53 us in situ, tools/x3_probe.py)
      const int q = tid + 256 * (rep >> 1);
      const int kq = (q & 3) * 8 + (rep & 1) * 4;
      offa[rep] = (uint32_t)min(bm + (q >> 2), M - 1) * (uint32_t)lda + (uint32_t)kq;
      ka[rep] = kq;
    } else {
      offa[rep] = (uint32_t)((tid & 7) * 4 + rep) * (uint32_t)lda + (uint32_t)min(bm + (tid >> 3) * 4, M - 4);
      ka[rep] = (tid & 7) * 4 + rep;
    }
  }
#pragma unroll
  for (int rep = 0; rep < 4; ++rep) {
    if (B_KCONTIG) {
      const int q = tid + 256 * (rep >> 1);
      const int kq = (q & 3) * 8 + (rep & 1) * 4;
      offb[rep] = (uint32_t)min(bn + (q >> 2), N - 1) * (uint32_t)ldb + (uint32_t)kq;
      kb4[rep] = kq;
    } else {
      offb[rep] = (uint32_t)((tid & 7) * 4 + rep) * (uint32_t)ldb + (uint32_t)min(bn + (tid >> 3) * 4, N - 4);
      kb4[rep] = (tid & 7) * 4 + rep;
    }
  }
  // One quad = the four elements that become two packed dwords per plane:
  //   k-contiguous operand: quad `rep` = one 16-B load (elements rep*4 .. rep*4+3);
  //   k-strided operand:    quad `e`   = column e of the 4x4 block (elements i*4+e, i = k inside the block).
  // The quads of A come first (NA4 of them), then the four of B.
  constexpr int NQ = NA4 + 4;
  auto quad_src = [&](int qd, int e) -> float& {
    if (qd < NA4) return A_KCONTIG ? xa[qd * 4 + e] : xa[e * 4 + qd];
    return B_KCONTIG ? xb[(qd - NA4) * 4 + e] : xb[e * 4 + (qd - NA4)];
  };
  // k (relative to the tile start) of element e of quad qd
  auto quad_k = [&](int qd, int e) -> int {
    if (qd < NA4) return A_KCONTIG ? ka[qd] : ka[e];
    return B_KCONTIG ? kb4[qd - NA4] : kb4[e];
  };
  auto quad_dst = [&](int qd, int p, int d) -> uint32_t& {   // d = 0/1: first / second dword of the quad in plane p
    if (qd < NA4) return A_KCONTIG ? pa[p * 2 * NA4 + qd * 2 + d] : pa[p * 8 + qd * 2 + d];
    return B_KCONTIG ? pb[p * 8 + (qd - NA4) * 2 + d] : pb[p * 8 + (qd - NA4) * 2 + d];
  };
  // issue the 16-B load `rep` of operand A (which = 0) / B (which = 1) for the tile starting at k0
  auto load_one = [&](int which, int rep, int k0, auto kcheck) {
    constexpr bool KCHECK = decltype(kcheck)::value;
    const bool kc = which == 0 ? A_KCONTIG : B_KCONTIG;
    const float* X = which == 0 ? A : Bm;
    const int ld = which == 0 ? lda : ldb;
    const uint32_t off = which == 0 ? offa[rep] : offb[rep];
    const int krel = which == 0 ? ka[rep] : kb4[rep];
    const float* src = X + (kc ? (int64_t)k0 : (int64_t)k0 * ld) + off;
    if (KCHECK) src = (k0 + krel < kend) ? src : X;      // past the K end: any valid address (zeroed at split)
#if defined(X3_PROBE_NOLOADA) || defined(X3_PROBE_NOLOADB)   // profiling aids: drop one operand's global loads
#ifdef X3_PROBE_NOLOADA
    if (which == 0) return;
#endif
#ifdef X3_PROBE_NOLOADB
    if (which == 1) return;
#endif
#endif
    const float4 v = ld4(src);
    float* dst = which == 0 ? &xa[rep * 4] : &xb[rep * 4];
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
  };
  auto load_all = [&](int k0, auto kcheck) {
#pragma unroll
    for (int rep = 0; rep < NA4; ++rep) load_one(0, rep, k0, kcheck);
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) load_one(1, rep, k0, kcheck);
  };
  uint32_t sh[4], sm[4], sl[4];   // split pieces of the quad in progress
  // micro-op m of the split of the tile starting at k1 (m = quad * 5 + step; steps 0-3 split one element,
  // step 4 packs); after a quad's elements are split its staging registers are refilled from tile k2
  auto split_micro = [&](int m, int k1, int k2, auto kcheck) {
    constexpr bool KCHECK = decltype(kcheck)::value;
    const int qd = m / 5, step = m % 5;
    if (step < 4) {
      float x = quad_src(qd, step);
      if (KCHECK) x = (k1 + quad_k(qd, step) < kend) ? x : 0.f;
#ifdef X3_PROBE_NOSPLIT          // profiling aid (tools/x3_probe.py): wrong numbers, same data movement
      sh[step] = sm[step] = sl[step] = __float_as_uint(x);
#else
      split3(x, sh[step], sm[step], sl[step]);
#endif
      if (step == 3) {
        const bool is_a = qd < NA4;
        const bool kc = is_a ? A_KCONTIG : B_KCONTIG;
        if (kc) {
          load_one(is_a ? 0 : 1, is_a ? qd : qd - NA4, k2, kcheck);
        } else if (qd == (is_a ? NA4 - 1 : NQ - 1)) {      // a 4x4 block is free once its last column is split
#pragma unroll
          for (int rep = 0; rep < 4; ++rep) load_one(is_a ? 0 : 1, rep, k2, kcheck);
        }
      }
    } else {
      quad_dst(qd, 0, 0) = pack2(sh[0], sh[1]); quad_dst(qd, 0, 1) = pack2(sh[2], sh[3]);
      quad_dst(qd, 1, 0) = pack2(sm[0], sm[1]); quad_dst(qd, 1, 1) = pack2(sm[2], sm[3]);
      quad_dst(qd, 2, 0) = pack2(sl[0], sl[1]); quad_dst(qd, 2, 1) = pack2(sl[2], sl[3]);
    }
  };
  auto write_tiles = [&]() {
#ifdef X3_PROBE_NOLDSW           // profiling aid: no LDS writes (keeps the packed registers live)
    if (pa[0] != 0x12345678u || pb[0] != 0x12345678u) return;
#endif
    if (A_KCONTIG) {
#pragma unroll
      for (int pr = 0; pr < NA4 / 2; ++pr) {
        const int q = tid + 256 * pr;
#pragma unroll
        for (int p = 0; p < 3; ++p)
          *reinterpret_cast<uint4*>(&Ap[p][(q >> 2) * TLD + (q & 3) * 8]) =
              make_uint4(pa[p * 2 * NA4 + pr * 4], pa[p * 2 * NA4 + pr * 4 + 1], pa[p * 2 * NA4 + pr * 4 + 2],
                         pa[p * 2 * NA4 + pr * 4 + 3]);
      }
    } else if (a_mine) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          *reinterpret_cast<uint2*>(&Ap[p][((tid >> 3) * 4 + e) * TLD + (tid & 7) * 4]) =
              make_uint2(pa[p * 8 + e * 2], pa[p * 8 + e * 2 + 1]);
    }
    if (B_KCONTIG) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int q = tid + 256 * pr;
#pragma unroll
        for (int p = 0; p < 3; ++p)
          *reinterpret_cast<uint4*>(&Bp[p][(q >> 2) * TLD + (q & 3) * 8]) =
              make_uint4(pb[p * 8 + pr * 4], pb[p * 8 + pr * 4 + 1], pb[p * 8 + pr * 4 + 2], pb[p * 8 + pr * 4 + 3]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          *reinterpret_cast<uint2*>(&Bp[p][((tid >> 3) * 4 + e) * TLD + (tid & 7) * 4]) =
              make_uint2(pb[p * 8 + e * 2], pb[p * 8 + e * 2 + 1]);
    }
  };

  f32x16 acc[WM][2];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int r31 = lane & 31, kh = lane >> 5;
  constexpr int KS = TBK / 16;                  // MFMA k-steps per tile
  constexpr int NM1 = 6 * WM * 2;               // MFMAs per k-step (24 / 12)
  constexpr int NM = KS * NM1;                  // per tile (48 / 24)
  constexpr int NF1 = 3 * (WM + 2);             // fragment reads per k-step (12 / 9)
  constexpr int MU = NQ * 5;                    // split micro-ops per tile
  bf16x8 af[KS][WM][3], bf[KS][2][3];
  auto read_frag = [&](int ks, int f) {         // f: p-major, A tiles then B tiles
    const int p = f / (WM + 2), i = f % (WM + 2);
#ifdef X3_PROBE_NOLDSR           // profiling aid: fragments from registers instead of LDS
    {
      uint4 z = make_uint4(pa[0], pa[1], pb[0], pb[1] + (uint32_t)f);
      if (i < WM) af[ks][i][p] = __builtin_bit_cast(bf16x8, z); else bf[ks][i - WM][p] = __builtin_bit_cast(bf16x8, z);
      return;
    }
#endif
    if (i < WM) {
      const uint4 va = *reinterpret_cast<const uint4*>(&Ap[p][(wm * 32 * WM + i * 32 + r31) * TLD + ks * 16 + kh * 8]);
      af[ks][i][p] = __builtin_bit_cast(bf16x8, va);
    } else {
      const uint4 vb = *reinterpret_cast<const uint4*>(&Bp[p][(wn * 64 + (i - WM) * 32 + r31) * TLD + ks * 16 + kh * 8]);
      bf[ks][i - WM][p] = __builtin_bit_cast(bf16x8, vb);
    }
  };
  // One tile: the MFMAs of the tile resident in LDS, in program order slot by slot; every slot also carries
  // its share of the next tile's split (VALU in the shadow of the 32-cycle MFMA), of the tile-after-next's
  // global loads and of the second k-step's fragment reads.  sched_barrier pins that interleave.
  // Terms smallest first: (1,1) (0,2) (2,0) (0,1) (1,0) (0,0); term-major, so neighbouring MFMAs hit
  // different accumulators.
  auto phase = [&](int k0, auto kcheck) {
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int f = 0; f < NF1; ++f) read_frag(0, f);
#ifdef X3_SETPRIO      // tools/x3_probe.py: the matrix phase of this wave ahead of the co-resident block's split / LDS-write
    __builtin_amdgcn_s_setprio(X3_SETPRIO);      // phase.  In isolation 53.3 -> 50.0 us (projection), 70.2 -> 68.2 (X^T dY); in the step nothing
#endif                 // (cfg-3 1.4784 vs 1.4852, cfg-5 20.47 vs 20.64 ms): the side streams' kernels pay it back.  Off.
#pragma unroll
    for (int g = 0; g < NM; ++g) {
      const int ks = g / NM1, t = (g % NM1) / (WM * 2), i = (g / 2) % WM, j = g % 2;
      if (KS > 1 && g >= NM1 - NF1 - 2 && g < NM1 - 2) read_frag(1, g - (NM1 - NF1 - 2));   // lands before k-step 1
#ifdef X3_PROBE_NOMFMA           // profiling aid: everything but the matrix instruction
      acc[i][j][g % 16] += (float)af[ks][i][TA[t]][0] * (float)bf[ks][j][TB[t]][0];
#else
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i][TA[t]], bf[ks][j][TB[t]], acc[i][j], 0, 0, 0);
#endif
#pragma unroll
      for (int m = g * MU / NM; m < (g + 1) * MU / NM; ++m) split_micro(m, k0 + TBK, k0 + 2 * TBK, kcheck);
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef X3_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };

  // tiles [0, nfull) are whole; the main loop only ever splits and loads whole tiles (no bounds work at
  // all), the last iterations (which reach the K tail or run past kend) take the checked variant
  const int nfull = (kend - kbeg) / TBK;
  load_all(kbeg, std::true_type());
#pragma unroll
  for (int m = 0; m < MU; ++m) {               // split tile 0 (and fetch tile 1 behind it)
    const int qd = m / 5, step = m % 5;
    (void)qd; (void)step;
    split_micro(m, kbeg, kbeg + TBK, std::true_type());
  }
  int k0 = kbeg;
  for (int t = 0; t + 2 < nfull; ++t, k0 += TBK) {
    write_tiles();                        // tile k0 (split during the previous phase)
    __syncthreads();
    phase(k0, std::false_type());
    __syncthreads();
  }
  for (; k0 < kend; k0 += TBK) {
    write_tiles();
    __syncthreads();
    phase(k0, std::true_type());
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = bn + wn * 64 + j * 32 + r31;
      if (col >= N) continue;
      // What the epilogue reads per element (bias, the relu-gradient mask, the accumulate target) is fetched for the 16
      // rows of the sub-tile TOGETHER, from clamped rows, before anything is consumed: one element at a time every
      // load sat behind its own branch with a vmcnt(0) wait -- 64 dependent (cache-hit) round trips per lane and tile.
      const float* bsrc = pr.bias ? pr.bias : bias;
      const int bg = flags >> 16;
      float eb[16], ey[16], ec[16];
      const bool has_b = !slab && (flags & XF_BIAS), has_y = !slab && (flags & XF_RELUGRAD), has_c = !slab && (flags & XF_ACC);
      int rows[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rows[r] = min(bm + wm * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh, M - 1);
      if (has_b) {
        if (bg) {
#pragma unroll
          for (int r = 0; r < 16; ++r) eb[r] = bsrc[(int64_t)(rows[r] / bg) * N + col];
        } else {
          const float b0 = bsrc[col];
#pragma unroll
          for (int r = 0; r < 16; ++r) eb[r] = b0;
        }
      }
      if (has_y) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ey[r] = reinterpret_cast<const float*>(mask)[(int64_t)rows[r] * N + col];
      }
      if (has_c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ec[r] = C[(int64_t)rows[r] * ldc + col];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = bm + wm * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row >= M) continue;
#ifdef X3_PROBE_NOSTORE          // profiling aid: keep the result live, store (almost) nothing
        if (acc[i][j][r] != 12345.678f) continue;
#endif
        if (slab) {
          slab[((int64_t)bz * M + row) * N + col] = acc[i][j][r];
        } else {
          float v = acc[i][j][r];
          if (has_b) v += eb[r];
          if (flags & XF_RELU) v = fmaxf(v, 0.f);
          if (flags & XF_DROP) {
            const uint64_t e = (uint64_t)row * (uint64_t)N + (uint64_t)col;
            const bool on = mask ? (mask[e] != 0) : (hash_uniform(seed, e) < keep);
            v = on ? v / keep : 0.f;
          }
          if (has_y) v = ey[r] > 0.f ? v / keep : 0.f;
          C[(int64_t)row * ldc + col] = has_c ? ec[r] + v : v;
        }
      }
    }
}

int score_launch_gemm_bf16x3(int trans, int wm, const GemmGroup& g, const float* bias, int flags, float keep,
                             const uint8_t* mask, uint64_t seed, hipStream_t s) {
#define LX(TR, WMv)                                                                                              \
  hipLaunchKernelGGL((gemm_bf16x3_kernel<TR, WMv>), dim3(g.total_blocks), dim3(256), 0, s, g, bias, flags, keep, mask, \
                     seed)
  if (wm == 2) {
    if (trans == 0) LX(0, 2);
    else if (trans == 1) LX(1, 2);
    else LX(2, 2);
  } else {
    if (trans == 0) LX(0, 1);
    else if (trans == 1) LX(1, 1);
    else LX(2, 1);
  }
#undef LX
  SCORE_CHECK_LAUNCH();
  return 0;
}
