// fp32-accurate GEMM on the bf16 matrix cores ("bf16x3").
//
// gfx950's f32-input MFMA runs at the VALU rate (64 FLOP/clk/SIMD = 1/16 of bf16) and there is no
// xf32.  An fp32 value splits EXACTLY into three bf16 pieces (8+8+8 significand bits, by truncation):
// x = x0 + x1 + x2.  x*y = sum_{p,q} x_p y_q; every bf16 product is exact in fp32, and the three
// dropped terms (x1 y2, x2 y1, x2 y2) are below 2^-24 |x y| -- one fp32 rounding.  So six
// v_mfma_f32_32x32x16_bf16 per k-step reproduce an fp32 GEMM to fp32 accuracy at 16/6 = 2.7x the
// f32-MFMA rate.  The split is done once per element while staging the tile into LDS.
//
// Tile: 128x128x32 per 256-thread block, 2x2 waves of 64x64 (2x2 MFMA tiles each), LDS holds the
// three bf16 planes of A as [m][k] and of B as [n][k] (k contiguous, 16-B fragments, row stride 40
// bf16 = 80 B: conflict-free ds_read_b128).  Same operand layouts / epilogue / split-K as gemm.hip.
#include <type_traits>
#include "common.h"
#include "kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TBN 128
#define TBK 32
#define TLD 40   // bf16 per LDS row (32 + 8 pad)

enum { XF_BIAS = 1, XF_RELU = 2, XF_ACC = 4, XF_DROP = 8 };

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
  l = __float_as_uint(r2) & 0xFFFF0000u;
}
// pack two bf16 (given as fp32-word upper halves) into one dword: lo element first
__device__ __forceinline__ uint32_t pack2(uint32_t a, uint32_t b) { return (a >> 16) | b; }

// WM = 32-row MFMA tiles per wave along M: block tile (64*WM) x 128
template <int TRANS, int WM>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                          const float* __restrict__ Bm, int ldb,
                                                          float* __restrict__ C, int ldc,
                                                          const float* __restrict__ bias, int flags, float keep,
                                                          const uint8_t* __restrict__ mask, uint64_t seed,
                                                          int k_chunk, float* __restrict__ slab) {
  constexpr int TBM = 64 * WM;
  constexpr int EA = TBM * TBK / 256;      // A elements staged per thread (16 or 8)
  __shared__ __attribute__((aligned(16))) unsigned short Ap[3][TBM * TLD];
  __shared__ __attribute__((aligned(16))) unsigned short Bp[3][TBN * TLD];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int bm = blockIdx.y * TBM, bn = blockIdx.x * TBN;
  const int kbeg = blockIdx.z * k_chunk;
  const int kend = min(K, kbeg + k_chunk);
  constexpr bool A_KCONTIG = (TRANS != 2);   // A[m][k]
  constexpr bool B_KCONTIG = (TRANS == 1);   // B[n][k]

  // Register pipeline: tile t is in LDS, tile t+1 is being split (VALU, interleaved with the MFMAs of
  // tile t), tile t+2 goes in flight from global memory as soon as the split has consumed the registers.  All staging is branch-free: addresses are
  // clamped into range and out-of-range elements are zeroed by a select.
  constexpr int XA = A_KCONTIG ? EA : 16;  // a k-strided operand is staged as 4x4 blocks: 16 per active thread
  float xa[XA], xb[16];                    // staging registers (tile t+1, then t+2)
  uint32_t pa[3 * XA / 2], pb[24];
  // k-contiguous operand X[r][k] (r = m or n): thread -> 4 quads, quad q: row q>>3, k (q&7)*4
  auto load_kc = [&](const float* X, int ld, int r0, int rlim, int k0, float* dst, auto nrep) {
#pragma unroll
    for (int rep = 0; rep < decltype(nrep)::value; ++rep) {
      const int q = tid + 256 * rep;
      const int r = r0 + (q >> 3), k = k0 + (q & 7) * 4;
      const bool ok = r < rlim && k < kend;
      const float4 v = ld4(X + (int64_t)(ok ? r : 0) * ld + (ok ? k : 0));
      dst[rep * 4 + 0] = ok ? v.x : 0.f; dst[rep * 4 + 1] = ok ? v.y : 0.f;
      dst[rep * 4 + 2] = ok ? v.z : 0.f; dst[rep * 4 + 3] = ok ? v.w : 0.f;
    }
  };
  // k-strided operand X[k][c] (c = m or n): thread -> a 4(k) x 4(c) block: k-group tid&7, column-group
  // tid>>3; four 16-B loads along c (one per k), transposed in registers on the way to LDS
  auto load_ks = [&](const float* X, int ld, int c0, int clim, int k0, float* dst, int ncols) {
    const int c = c0 + (tid >> 3) * 4;
    const int kb = k0 + (tid & 7) * 4;
    const bool mine = (tid >> 3) * 4 < ncols;      // a 64-wide tile has only 128 blocks of 4x4
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = mine && c < clim && kb + i < kend;
      const float4 v = ld4(X + (int64_t)(ok ? kb + i : 0) * ld + (ok ? c : 0));
      dst[i * 4 + 0] = ok ? v.x : 0.f; dst[i * 4 + 1] = ok ? v.y : 0.f;
      dst[i * 4 + 2] = ok ? v.z : 0.f; dst[i * 4 + 3] = ok ? v.w : 0.f;
    }
  };
  // split 16 staged floats into packed bf16 planes: dst[p*8 + d], d = dword index inside the plane
  // n floats -> planes dst[p*(n/2) + d]
  auto split16 = [&](const float* src, uint32_t* dst, auto nn) {
    constexpr int NH = decltype(nn)::value / 2;
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      uint32_t h0, m0, l0, h1, m1, l1;
      split3(src[2 * j], h0, m0, l0);
      split3(src[2 * j + 1], h1, m1, l1);
      dst[j] = pack2(h0, h1); dst[NH + j] = pack2(m0, m1); dst[2 * NH + j] = pack2(l0, l1);
    }
  };
  auto write_kc = [&](unsigned short* P, int plane_stride, const uint32_t* src, auto nrep) {
    constexpr int NR = decltype(nrep)::value;
#pragma unroll
    for (int rep = 0; rep < NR; ++rep) {
      const int q = tid + 256 * rep;
      const int r = q >> 3, k = (q & 7) * 4;
#pragma unroll
      for (int p = 0; p < 3; ++p)
        *reinterpret_cast<uint2*>(&P[p * plane_stride + r * TLD + k]) =
            make_uint2(src[p * 2 * NR + rep * 2], src[p * 2 * NR + rep * 2 + 1]);
    }
  };
  // (ks) the 4x4 block transposed: for column e the four k values are src[0*4+e] .. src[3*4+e]
  auto split16t = [&](const float* src, uint32_t* dst) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t h[4], m[4], l[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) split3(src[i * 4 + e], h[i], m[i], l[i]);
      dst[e * 2] = pack2(h[0], h[1]); dst[e * 2 + 1] = pack2(h[2], h[3]);
      dst[8 + e * 2] = pack2(m[0], m[1]); dst[8 + e * 2 + 1] = pack2(m[2], m[3]);
      dst[16 + e * 2] = pack2(l[0], l[1]); dst[16 + e * 2 + 1] = pack2(l[2], l[3]);
    }
  };
  auto write_ks = [&](unsigned short* P, int plane_stride, const uint32_t* src, int ncols) {
    const int c = (tid >> 3) * 4, kb = (tid & 7) * 4;
    if (c >= ncols) return;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        *reinterpret_cast<uint2*>(&P[p * plane_stride + (c + e) * TLD + kb]) =
            make_uint2(src[p * 8 + e * 2], src[p * 8 + e * 2 + 1]);
  };
  // (the k-strided A of a 64-row tile still stages 16 floats per ACTIVE thread: EA counts kc quads only)
  auto split_tiles = [&](const float* ra, const float* rb) {
    if (A_KCONTIG) split16(ra, pa, std::integral_constant<int, EA>()); else split16t(ra, pa);
    if (B_KCONTIG) split16(rb, pb, std::integral_constant<int, 16>()); else split16t(rb, pb);
  };
  auto load_tiles = [&](int k0, float* ra, float* rb) {
    if (A_KCONTIG) load_kc(A, lda, bm, M, k0, ra, std::integral_constant<int, EA / 4>()); else load_ks(A, lda, bm, M, k0, ra, TBM);
    if (B_KCONTIG) load_kc(Bm, ldb, bn, N, k0, rb, std::integral_constant<int, 4>()); else load_ks(Bm, ldb, bn, N, k0, rb, TBN);
  };
  auto write_tiles = [&]() {
    if (A_KCONTIG) write_kc(&Ap[0][0], TBM * TLD, pa, std::integral_constant<int, EA / 4>()); else write_ks(&Ap[0][0], TBM * TLD, pa, TBM);
    if (B_KCONTIG) write_kc(&Bp[0][0], TBN * TLD, pb, std::integral_constant<int, 4>()); else write_ks(&Bp[0][0], TBN * TLD, pb, TBN);
  };

  f32x16 acc[WM][2];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int r31 = lane & 31, kh = lane >> 5;
  // MFMA phase on the tile resident in LDS, with the next tile's split (VALU) spread into the MFMA
  // issue gaps and the loads of the tile after that issued first
  auto mfma_phase = [&]() {
#pragma unroll
    for (int ks = 0; ks < TBK / 16; ++ks) {
      bf16x8 af[WM][3], bf[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          const uint4 va = *reinterpret_cast<const uint4*>(&Ap[p][(wm * 32 * WM + i * 32 + r31) * TLD + ks * 16 + kh * 8]);
          af[i][p] = __builtin_bit_cast(bf16x8, va);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint4 vb = *reinterpret_cast<const uint4*>(&Bp[p][(wn * 64 + i * 32 + r31) * TLD + ks * 16 + kh * 8]);
          bf[i][p] = __builtin_bit_cast(bf16x8, vb);
        }
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 c = acc[i][j];
          // smallest terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
#pragma unroll
    for (int g = 0; g < 24 * WM; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA  (32 cycles of matrix pipe, 8 of issue)
      __builtin_amdgcn_sched_group_barrier(0x002, WM == 2 ? 5 : 8, 0);   // VALU of the next tile's split
    }
  };
  load_tiles(kbeg, xa, xb);
  split_tiles(xa, xb);
  load_tiles(kbeg + TBK, xa, xb);         // (zero-filled past kend)
  for (int k0 = kbeg; k0 < kend; k0 += TBK) {
    write_tiles();                        // tile k0
    __syncthreads();
    split_tiles(xa, xb);                  // tile k0+1 (its loads were issued one phase ago)
    load_tiles(k0 + 2 * TBK, xa, xb);     // tile k0+2: issued as soon as the split has read xa/xb
    mfma_phase();
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = bn + wn * 64 + j * 32 + r31;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = bm + wm * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row >= M) continue;
        if (slab) {
          slab[((int64_t)blockIdx.z * M + row) * N + col] = acc[i][j][r];
        } else {
          float v = x3_epilogue(acc[i][j][r], row, col, N, bias, flags, keep, mask, seed);
          float* dst = C + (int64_t)row * ldc + col;
          *dst = (flags & XF_ACC) ? *dst + v : v;
        }
      }
    }
}

int score_launch_gemm_bf16x3(int trans, int wm, dim3 grid, int M, int N, int K, const float* A, int lda,
                             const float* Bm, int ldb, float* C, int ldc, const float* bias, int flags, float keep,
                             const uint8_t* mask, uint64_t seed, int k_chunk, float* slab, hipStream_t s) {
#define LX(TR, WMv)                                                                                               \
  hipLaunchKernelGGL((gemm_bf16x3_kernel<TR, WMv>), grid, dim3(256), 0, s, M, N, K, A, lda, Bm, ldb, C, ldc, bias, \
                     flags, keep, mask, seed, k_chunk, slab)
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
