// fp32-accurate GEMM against a WEIGHT matrix on the bf16 matrix cores, register-only ("x3w").
//
// Same arithmetic as gemm_bf16x3.hip (every fp32 operand split exactly into three bf16 planes, six
// v_mfma_f32_32x32x16_bf16 per k-step, the three terms below 2^-24 dropped, fp32 accumulation), different data path.
// There, both operands are split on the fly and staged through LDS, and the LDS *write* path (64-85 B/clk per CU for
// six bytes per element) is what the matrix pipe waits for (tools/x3_probe.py: pipe busy 41 %).  When one operand
// is a weight matrix -- the GRU input projections x.[Wx_gates|Wx_cand] and their d x = d(xproj).Wx^T, the two
// largest products of the path -- it is the same for every row block and changes once per step, so it is split ONCE
// per step into MFMA fragment order (x3w_frag_kernel): a wave then takes the B operand of one k-step and 32-column
// tile with coalesced 16-byte loads straight from L2 into registers, the way gru_stream.hip streams its recurrent
// weights.  The activation operand A[m][k] (k contiguous) is read by the wave that owns those rows, 32 bytes per
// lane and k-step, and split in registers in the shadow of the MFMAs.  No LDS, no barriers: the waves of a workgroup
// only share cache lines (2 x 2 waves: each A line and each B fragment is used by two of them).
//
// MEASURED (tools/x3w_probe.py, tools/x3w_strip.py; MI355X): correct to the same bound as the other GEMM kernels, but
// SLOWER than the LDS-staged kernel -- [20480,448]x[448,384]: 66 us (+6 us fragment build) vs 60 us; cfg-5's
// [196608,384]x[384,768]: 0.99 vs 0.82 ms.  With both operands' loads stripped it runs 41 us (MFMA + 31 MB of stores),
// each operand's loads add 12-15 us and a deeper A ring makes it worse: every wave pulls 10 KB per k-step through
// the CU's vector L1 (64 B/clk: 8 waves x 320 L1-cycles per 1,536-cycle step window), which saturates before the
// matrix pipe does.  Sharing operands across waves needs LDS -- the existing kernel.  score_forward / score_backward
// therefore do NOT use this kernel; it stays as an op-level entry point (score_gemm_weights) with its test and
// probes so the next attempt starts from the measurement.
//
// k is dealt so that a 128-byte line of A is consumed whole by two consecutive k-steps: step st = 2*kt + s covers
// k = 32*kt + 16*kh + 8*s + j (kh = lane >> 5, j = 0..7) -- the fragment builder uses the same map.
#include "../../score_amd/csrc/common.h"
#include "../../score_amd/csrc/kernels.h"
// (round 3: out of the product library -- measured slower than the LDS-staged kernel, the step never used it; kept here with
//  its probes, tools/x3w_probe.py and tools/x3w_strip.py, which build it through tools/x3w_wrap.hip)
bool score_x3w_ok(int M, int N, int K, int lda);
int64_t score_x3w_frag_floats(int N, int K);
int score_launch_x3w_frag(const float* W, int ldw, int K, int N, int trans, float* out, hipStream_t s);
int score_launch_gemm_x3w(int n, int M, int N, int K, const float* const* A, int lda, const float* const* Bfrag,
                          float* const* C, int ldc, const float* const* bias, int flags, hipStream_t s);
extern "C" int score_gemm_weights(int32_t trans, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* W,
                                  int32_t ldw, float* C, int32_t ldc, const float* bias, int32_t flags, float* scratch,
                                  int64_t scratch_floats, void* stream);
extern "C" int64_t score_gemm_weights_scratch_floats(int32_t N, int32_t K);

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define XW_MT 2          // 32-row MFMA tiles per wave
#define XW_NT 2          // 32-column MFMA tiles per wave
#define XW_WM 2          // waves per workgroup along M
#define XW_WN 2          // ... along N
#define XW_TBM (32 * XW_MT * XW_WM)
#define XW_TBN (32 * XW_NT * XW_WN)
#ifndef XW_RA
#define XW_RA 4          // k-steps of A in flight per wave
#endif
#define XW_MAXP 4        // problems of one shape per launch (both GRU sides)

enum { XWF_BIAS = 1, XWF_RELU = 2 };

__device__ __forceinline__ void xw_split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  const uint32_t xb = __float_as_uint(x);
  h = xb & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xFFFF0000u;
  const float r2 = r1 - __uint_as_float(m);
  l = __float_as_uint(r2);       // (xw_pack2 keeps the upper half only)
}
__device__ __forceinline__ uint32_t xw_pack2(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// out[((nt * S + st) * 3 + plane) * 64 + lane] (16 B each) = the eight bf16 of plane `plane` of
// B(k, n), n = 32*nt + (lane & 31), k = 32*(st >> 1) + 16*(lane >> 5) + 8*(st & 1) + j;  S = K / 16.
// B(k, n) = W[k*ldw + n] (trans 0) or W[n*ldw + k] (trans 1); columns n >= N are zero.
__global__ void x3w_frag_kernel(const float* __restrict__ W, int ldw, int K, int N, int trans, uint4* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int S = K / 16, NT = (N + 31) / 32;
  if (i >= (int64_t)NT * S * 64) return;
  const int lane = (int)(i & 63), st = (int)((i >> 6) % S), nt = (int)((i >> 6) / S);
  const int n = nt * 32 + (lane & 31);
  const int k0 = 32 * (st >> 1) + 16 * (lane >> 5) + 8 * (st & 1);
  uint32_t h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float x = 0.f;
    if (n < N) x = trans ? W[(int64_t)n * ldw + k0 + j] : W[(int64_t)(k0 + j) * ldw + n];
    xw_split3(x, h[j], m[j], l[j]);
  }
  uint4* o = out + ((int64_t)(nt * S + st) * 3) * 64 + lane;
  o[0] = make_uint4(xw_pack2(h[0], h[1]), xw_pack2(h[2], h[3]), xw_pack2(h[4], h[5]), xw_pack2(h[6], h[7]));
  o[64] = make_uint4(xw_pack2(m[0], m[1]), xw_pack2(m[2], m[3]), xw_pack2(m[4], m[5]), xw_pack2(m[6], m[7]));
  o[128] = make_uint4(xw_pack2(l[0], l[1]), xw_pack2(l[2], l[3]), xw_pack2(l[4], l[5]), xw_pack2(l[6], l[7]));
}

struct XwProb { const float* A; const uint4* Bf; float* C; const float* bias; };
struct XwArgs {
  XwProb p[XW_MAXP];
  int n, M, N, K, lda, ldc, flags, gx, gy, blocks_per_problem;   // flags: XWF_* | (bias row group << 16)
};

__global__ __launch_bounds__(64 * XW_WM * XW_WN, 2) void gemm_x3w_kernel(const XwArgs a) {
  const int pi = (int)blockIdx.x / a.blocks_per_problem;
  const int local = (int)blockIdx.x - pi * a.blocks_per_problem;
  if (local >= a.gx * a.gy) return;
  const XwProb& pr = a.p[pi];
  int bx, by, bz;
  xcd_tile_coords_n(a.gx * a.gy, local, a.gx, a.gy, bx, by, bz);
  const int M = a.M, N = a.N, K = a.K, lda = a.lda;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / XW_WN, wn = wave % XW_WN;
  const int li = lane & 31, kh = lane >> 5;
  const int row0 = by * XW_TBM + wm * 32 * XW_MT;      // first row / column of this wave's tile
  const int col0 = bx * XW_TBN + wn * 32 * XW_NT;
  const int S = K / 16;
  const int NTp = (N + 31) / 32;

  // A: this lane's row of every m-tile (clamped: rows past M are computed and never stored)
  const float* ap[XW_MT];
#pragma unroll
  for (int mt = 0; mt < XW_MT; ++mt)
    ap[mt] = pr.A + (int64_t)min(row0 + mt * 32 + li, M - 1) * lda + kh * 16;
  // B fragments of this wave's column tiles (a tile past the padded width reads tile 0 and is never stored)
  const uint4* bp[XW_NT];
#pragma unroll
  for (int nt = 0; nt < XW_NT; ++nt) {
    const int t = col0 / 32 + nt;
    bp[nt] = pr.Bf + (int64_t)(t < NTp ? t : 0) * S * 3 * 64 + lane;
  }

  f32x16 acc[XW_MT][XW_NT];
#pragma unroll
  for (int i = 0; i < XW_MT; ++i)
#pragma unroll
    for (int j = 0; j < XW_NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // rings: A (activations: first touch comes from HBM / Infinity Cache) runs XW_RA k-steps ahead, B (weights: L2-resident)
  // two; slot = step % ring size
  constexpr int RA = XW_RA, RB = 2;
  float4 ra[RA][XW_MT][2];           // raw A: 8 floats per m-tile and step
  uint4 rb[RB][XW_NT][3];            // B fragments: 3 planes per column tile and step
  auto load_a = [&](int s, int st) {
    const int ko = 32 * (st >> 1) + 8 * (st & 1);
#pragma unroll
    for (int mt = 0; mt < XW_MT; ++mt) {
#ifdef XWP_NOLOADA               // tools/x3w_strip.py: one ingredient stripped at a time (timing only)
      ra[s][mt][0] = make_float4(1.f + ko, 2.f, 3.f, 4.f); ra[s][mt][1] = make_float4(5.f, 6.f, 7.f, 8.f + mt);
#else
      ra[s][mt][0] = ld4_global(ap[mt] + ko);
      ra[s][mt][1] = ld4_global(ap[mt] + ko + 4);
#endif
    }
  };
  auto load_b = [&](int s, int st) {
#pragma unroll
    for (int nt = 0; nt < XW_NT; ++nt)
#pragma unroll
      for (int p = 0; p < 3; ++p)
      {
#ifdef XWP_NOLOADB
        const float4 v = make_float4(1.f + st, 2.f + p, 3.f, 4.f + nt);
#else
        const float4 v = ld4_global(reinterpret_cast<const float*>(bp[nt] + ((int64_t)st * 3 + p) * 64));
#endif
        rb[s][nt][p] = make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w));
      }
  };
#pragma unroll
  for (int i = 0; i < RA; ++i) load_a(i, i);
#pragma unroll
  for (int i = 0; i < RB; ++i) load_b(i, i);

  // terms smallest first: (1,1) (0,2) (2,0) (0,1) (1,0) (0,0); term-major, so neighbouring MFMAs hit different accumulators
  constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll 1
  for (int s0 = 0; s0 < S; s0 += RA) {
#pragma unroll
    for (int u = 0; u < RA; ++u) {
      const int st = s0 + u;
      // split this step's A rows into the three planes (VALU, in the shadow of the previous step's MFMAs)
      bf16x8 af[XW_MT][3];
#pragma unroll
      for (int mt = 0; mt < XW_MT; ++mt) {
        const float x[8] = {ra[u][mt][0].x, ra[u][mt][0].y, ra[u][mt][0].z, ra[u][mt][0].w,
                            ra[u][mt][1].x, ra[u][mt][1].y, ra[u][mt][1].z, ra[u][mt][1].w};
        uint32_t h[8], m[8], l[8];
#pragma unroll
#ifdef XWP_NOSPLIT
        for (int j = 0; j < 8; ++j) h[j] = m[j] = l[j] = __float_as_uint(x[j]);
#else
        for (int j = 0; j < 8; ++j) xw_split3(x[j], h[j], m[j], l[j]);
#endif
        af[mt][0] = __builtin_bit_cast(bf16x8, make_uint4(xw_pack2(h[0], h[1]), xw_pack2(h[2], h[3]), xw_pack2(h[4], h[5]), xw_pack2(h[6], h[7])));
        af[mt][1] = __builtin_bit_cast(bf16x8, make_uint4(xw_pack2(m[0], m[1]), xw_pack2(m[2], m[3]), xw_pack2(m[4], m[5]), xw_pack2(m[6], m[7])));
        af[mt][2] = __builtin_bit_cast(bf16x8, make_uint4(xw_pack2(l[0], l[1]), xw_pack2(l[2], l[3]), xw_pack2(l[4], l[5]), xw_pack2(l[6], l[7])));
      }
      // the raw registers are free: step st + RA goes in flight (the last steps re-read the final ones: valid
      // addresses, unused)
      load_a(u, min(st + RA, S - RA + u));
      bf16x8 bfr[XW_NT][3];
#pragma unroll
      for (int nt = 0; nt < XW_NT; ++nt)
#pragma unroll
        for (int p = 0; p < 3; ++p) bfr[nt][p] = __builtin_bit_cast(bf16x8, rb[u % RB][nt][p]);
      load_b(u % RB, min(st + RB, S - RB + (u % RB)));
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int mt = 0; mt < XW_MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < XW_NT; ++nt)
#ifdef XWP_NOMFMA
            acc[mt][nt][(t * 2 + mt) & 15] += (float)af[mt][TA[t]][0] * (float)bfr[nt][TB[t]][0];
#else
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][TA[t]], bfr[nt][TB[t]], acc[mt][nt], 0, 0, 0);
#endif
    }
  }

  const int bgrp = a.flags >> 16;
#pragma unroll
  for (int mt = 0; mt < XW_MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < XW_NT; ++nt) {
      const int col = col0 + nt * 32 + li;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row >= M) continue;
        float v = acc[mt][nt][r];
        if (a.flags & XWF_BIAS) v += pr.bias[bgrp ? (int64_t)(row / bgrp) * N + col : col];
        if (a.flags & XWF_RELU) v = fmaxf(v, 0.f);
        pr.C[(int64_t)row * a.ldc + col] = v;
      }
    }
}

// ------------------------------------------------------------------------------------------------ host side
bool score_x3w_ok(int M, int N, int K, int lda) {
  // K in whole trips of two ring revolutions; rows 16-B aligned; enough rows for the 128-row workgroup tiles to pay
  return K >= 64 && (K % 64) == 0 && (lda % 4) == 0 && M >= 512 && N >= 32;
}
int64_t score_x3w_frag_floats(int N, int K) { return (int64_t)((N + 31) / 32) * (K / 16) * 3 * 64 * 4; }

int score_launch_x3w_frag(const float* W, int ldw, int K, int N, int trans, float* out, hipStream_t s) {
  if (!W || !out || K <= 0 || (K % 64) || N <= 0) return SCORE_E_SHAPE;
  if (trans && ((ldw & 3) || (reinterpret_cast<uintptr_t>(W) & 15))) return SCORE_E_SHAPE;
  const int64_t n = (int64_t)((N + 31) / 32) * (K / 16) * 64;
  hipLaunchKernelGGL(x3w_frag_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, W, ldw, K, N, trans,
                     reinterpret_cast<uint4*>(out));
  SCORE_CHECK_LAUNCH();
  return 0;
}

// C_i[M,N] = epi(A_i[M,K] . B_i), i < n problems of one shape; B_i given as fragments (score_launch_x3w_frag)
int score_launch_gemm_x3w(int n, int M, int N, int K, const float* const* A, int lda, const float* const* Bfrag,
                          float* const* C, int ldc, const float* const* bias, int flags, hipStream_t s) {
  if (n <= 0 || n > XW_MAXP || !score_x3w_ok(M, N, K, lda)) return SCORE_E_SHAPE;
  XwArgs a;
  a.n = n; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.flags = flags;
  a.gx = (N + XW_TBN - 1) / XW_TBN; a.gy = (M + XW_TBM - 1) / XW_TBM;
  a.blocks_per_problem = (a.gx * a.gy + 7) & ~7;       // (index inside the problem) % 8 still names the XCD group
  for (int i = 0; i < n; ++i) {
    if (!A[i] || !Bfrag[i] || !C[i] || ((flags & XWF_BIAS) && !(bias && bias[i]))) return SCORE_E_BADARG;
    if (reinterpret_cast<uintptr_t>(A[i]) & 15) return SCORE_E_SHAPE;
    a.p[i].A = A[i]; a.p[i].Bf = reinterpret_cast<const uint4*>(Bfrag[i]); a.p[i].C = C[i];
    a.p[i].bias = bias ? bias[i] : nullptr;
  }
  hipLaunchKernelGGL(gemm_x3w_kernel, dim3(n * a.blocks_per_problem), dim3(64 * XW_WM * XW_WN), 0, s, a);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------ C-ABI (op level)
extern "C" int64_t score_gemm_weights_scratch_floats(int32_t N, int32_t K) {
  return (K > 0 && (K % 64) == 0 && N > 0) ? score_x3w_frag_floats(N, K) : 0;
}
extern "C" int score_gemm_weights(int32_t trans, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda,
                                  const float* W, int32_t ldw, float* C, int32_t ldc, const float* bias, int32_t flags,
                                  float* scratch, int64_t scratch_floats, void* stream) {
  if (!A || !W || !C || !scratch || M <= 0 || N <= 0 || K <= 0) return SCORE_E_BADARG;
  if ((trans != 0 && trans != 1) || (flags & ~3)) return SCORE_E_BADARG;
  if (!score_x3w_ok(M, N, K, lda)) return SCORE_E_SHAPE;
  if (scratch_floats < score_x3w_frag_floats(N, K) || (reinterpret_cast<uintptr_t>(scratch) & 15)) return SCORE_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  SCORE_TRY(score_launch_x3w_frag(W, ldw, K, N, trans, scratch, s));
  const float* Ap[1] = {A};
  const float* Bp[1] = {scratch};
  float* Cp[1] = {C};
  const float* bp[1] = {bias};
  return score_launch_gemm_x3w(1, M, N, K, Ap, lda, Bp, Cp, ldc, bp, flags, s);
}
