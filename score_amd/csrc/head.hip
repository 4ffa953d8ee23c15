// Pointwise / small-reduction kernels of the temporal attention (score.py:169-186),
// pooled states (:214-215), build_fc_net (:68-76), build_logloss / build_l2norm
// (:78-94) and ApplyAdam (:96-99).  All HBM- or latency-bound; GEMMs live in gemm.hip.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

// The first attention layer acts on inp = [q, k, q-k, q*k] (score.py:173-174), k = [user_rep | item_rep |
// atten_info], q broadcast over the T slices.  With W1 = [Wa; Wb; Wc; Wd] (Dk rows each)
//   inp . W1 = q . (Wa + Wc)  +  k . (Wb - Wc)  +  (q*k) . Wd
// so the [B*T]-row product only needs [k, q*k] (2 Dk columns, half the bytes and flops of the literal
// form) against Weff = [Wb - Wc; Wd]; the q term is one [B, Dk] x [Dk, 80] product added per sample
// (score_gemm's row-grouped bias).
// `copies` replicas of Weff, `copy_stride` floats apart: the fused attention forward (head_fused.hip) has every workgroup
// read all of Weff at its start, and 32 CUs of an XCD asking one L2 channel for the same line at the same moment took
// 34 us for 189 KB; neighbouring workgroups read different replicas (different lines, different channels).
__device__ __forceinline__ void attn_fold_w1_body(int blk, int Dk, int NA, const float* __restrict__ W1, float* __restrict__ weff,
                                                  float* __restrict__ wq, int copies, int64_t copy_stride) {
  int i = blk * 256 + threadIdx.x;
  if (i >= Dk * NA) return;
  const float wa = W1[i], wb = W1[Dk * NA + i], wc = W1[2 * Dk * NA + i], wd = W1[3 * Dk * NA + i];
  for (int c = 0; c < copies; ++c) {
    weff[c * copy_stride + i] = wb - wc;
    weff[c * copy_stride + Dk * NA + i] = wd;
  }
  wq[i] = wa + wc;
}
__global__ __launch_bounds__(256) void attn_fold_w1_kernel(int Dk, int NA, const float* __restrict__ W1, float* __restrict__ weff,
                                                           float* __restrict__ wq, int copies, int64_t copy_stride) {
  attn_fold_w1_body(blockIdx.x, Dk, NA, W1, weff, wq, copies, copy_stride);
}

int score_launch_attn_fold_w1(int Dk, int NA, const float* W1, float* weff, float* wq, hipStream_t s, int copies,
                              int64_t copy_stride) {
  hipLaunchKernelGGL(attn_fold_w1_kernel, dim3((Dk * NA + 255) / 256), dim3(256), 0, s, Dk, NA, W1, weff, wq, copies,
                     copy_stride);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// inp2 = [k, q*k]
__global__ void attn_build_inp_kernel(int BT, int T, int H, int NI, const float* __restrict__ q,
                                      const float* __restrict__ ur, const float* __restrict__ ir,
                                      const float* __restrict__ info, float* __restrict__ inp) {
  const int Dk = 2 * H + NI;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)BT * Dk) return;
  int bt = (int)(i / Dk), j = (int)(i - (int64_t)bt * Dk);
  int b = bt / T;
  float k = j < H ? ur[(int64_t)bt * H + j]
                  : (j < 2 * H ? ir[(int64_t)bt * H + (j - H)] : info[(int64_t)bt * NI + (j - 2 * H)]);
  float qq = q[(int64_t)b * Dk + j];
  float* o = inp + (int64_t)bt * 2 * Dk;
  o[j] = k;
  o[Dk + j] = qq * k;
}
// the same, four columns per thread (H, NI multiples of 4): 16-B loads / stores, 32-bit index arithmetic, the
// source segment chosen by a pointer select instead of a branch around the load
__global__ __launch_bounds__(256) void attn_build_inp4_kernel(int BT, int T, int H, int NI, const float* __restrict__ q,
                                                              const float* __restrict__ ur, const float* __restrict__ ir,
                                                              const float* __restrict__ info, float* __restrict__ inp) {
  const int Dk = 2 * H + NI, Dk4 = Dk >> 2;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (unsigned)BT * (unsigned)Dk4) return;
  const unsigned bt = i / (unsigned)Dk4;
  const int j = (int)(i - bt * (unsigned)Dk4) * 4;
  const unsigned b = bt / (unsigned)T;
  const float* src = j < H ? ur + (int64_t)bt * H + j : (j < 2 * H ? ir + (int64_t)bt * H + (j - H) : info + (int64_t)bt * NI + (j - 2 * H));
  const float4 k = ld4(src);
  const float4 qq = ld4(q + (int64_t)b * Dk + j);
  float* o = inp + (int64_t)bt * 2 * Dk;
  st4(o + j, k);
  st4(o + Dk + j, make_float4(qq.x * k.x, qq.y * k.y, qq.z * k.z, qq.w * k.w));
}

int score_launch_attn_build_inp(int B, int T, int H, int NI, const float* q, const float* ur, const float* ir,
                                const float* info, float* inp, hipStream_t s) {
  int64_t n = (int64_t)B * T * (2 * H + NI);
  if ((H & 3) == 0 && (NI & 3) == 0 && n / 4 < (int64_t)1 << 31) {
    hipLaunchKernelGGL(attn_build_inp4_kernel, dim3((unsigned)cdiv64(n / 4, 256)), dim3(256), 0, s, B * T, T, H, NI, q, ur,
                       ir, info, inp);
    SCORE_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(attn_build_inp_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, B * T, T, H, NI, q,
                     ur, ir, info, inp);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// dzsum[b][n] = sum_t dz[b*T + t][n]  (t order): the gradient reaching the per-sample q term
__global__ void attn_dzsum_kernel(int B, int T, int NA, const float* __restrict__ dz, float* __restrict__ dzsum) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * NA) return;
  const int b = i / NA, n = i - b * NA;
  float acc = 0.f;
  for (int t = 0; t < T; ++t) acc += dz[((int64_t)b * T + t) * NA + n];
  dzsum[i] = acc;
}

int score_launch_attn_dzsum(int B, int T, int NA, const float* dz, float* dzsum, hipStream_t s) {
  hipLaunchKernelGGL(attn_dzsum_kernel, dim3((B * NA + 255) / 256), dim3(256), 0, s, B, T, NA, dz, dzsum);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// gradient of W1 from the folded pieces: dWa = dWq, dWb = dWeff_k, dWc = dWq - dWeff_k, dWd = dWeff_qk
__global__ void attn_w1_grad_kernel(int Dk, int NA, const float* __restrict__ dweff, const float* __restrict__ dwq,
                                    float* __restrict__ gW1) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Dk * NA) return;
  const float dq = dwq[i], dk = dweff[i], dqk = dweff[Dk * NA + i];
  gW1[i] = dq;
  gW1[Dk * NA + i] = dk;
  gW1[2 * Dk * NA + i] = dq - dk;
  gW1[3 * Dk * NA + i] = dqk;
}

int score_launch_attn_w1_grad(int Dk, int NA, const float* dweff, const float* dwq, float* gW1, hipStream_t s) {
  hipLaunchKernelGGL(attn_w1_grad_kernel, dim3((Dk * NA + 255) / 256), dim3(256), 0, s, Dk, NA, dweff, dwq, gW1);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// one wave per sample: fc3 = a2.w5 + b5, where(mask, fc3, -2^32+1), softmax over T,
// pooled states sum_t rep_t * score_t  (score.py:177-181, 214-215).  The T scores of a
// sample live in the wave's LDS row.
__global__ __launch_bounds__(256) void attn_pool_fwd_kernel(
    int B, int T, int H, int NA, const float* __restrict__ a2, const float* __restrict__ w5,
    const float* __restrict__ b5, const int32_t* __restrict__ length, const float* __restrict__ ur,
    const float* __restrict__ ir, float* __restrict__ score, float* __restrict__ head, int ldh, int off_u,
    int off_i) {
  extern __shared__ float sh[];  // [4][T]
  const int lane = threadIdx.x & 63;
  const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const bool bok = b < B;
  float* sc = sh + (threadIdx.x >> 6) * T;
  const int len = bok ? length[b] : 0;
  float mx = -INFINITY;
  for (int t = lane; t < T; t += 64) {
    float acc = 0.f;
    if (bok)
      for (int n = 0; n < NA; ++n) acc = fmaf(a2[((int64_t)b * T + t) * NA + n], w5[n], acc);
    float sv = t < len ? acc + b5[0] : -4294967295.0f;
    sc[t] = sv;
    mx = fmaxf(mx, sv);
  }
  mx = wave_max(mx);
  float den = 0.f;
  for (int t = lane; t < T; t += 64) {
    float e = expf(sc[t] - mx);
    sc[t] = e;
    den += e;
  }
  den = wave_sum(den);
  for (int t = lane; t < T; t += 64) {
    float v = sc[t] / den;
    sc[t] = v;
    if (bok) score[(int64_t)b * T + t] = v;
  }
  __syncthreads();
  if (!bok) return;
  for (int j = lane; j < H; j += 64) {
    float su = 0.f, si = 0.f;
    for (int t = 0; t < T; ++t) {
      float w = sc[t];
      su = fmaf(ur[((int64_t)b * T + t) * H + j], w, su);
      si = fmaf(ir[((int64_t)b * T + t) * H + j], w, si);
    }
    if (off_u >= 0) head[(int64_t)b * ldh + off_u + j] = su;
    if (off_i >= 0) head[(int64_t)b * ldh + off_i + j] = si;
  }
}

int score_launch_attn_pool_fwd(int B, int T, int H, int NA, const float* a2, const float* w5, const float* b5,
                               const int32_t* length, const float* ur, const float* ir, float* score, float* head,
                               int ldh, int off_u, int off_i, hipStream_t s) {
  hipLaunchKernelGGL(attn_pool_fwd_kernel, dim3((B + 3) / 4), dim3(256), 4 * T * sizeof(float), s, B, T, H, NA,
                     a2, w5, b5, length, ur, ir, score, head, ldh, off_u, off_i);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// The tail of the temporal attention in ONE launch, a block per sample (score.py:175-181, 214-215): dense_4
// (a1 [T, N1] -> relu -> a2 [T, N2]), dense_5 (-> 1), where(mask, ., -2^32+1), softmax over T and the pooled states.
// The sample's a1 rows and the layer's kernel sit in LDS; as two launches (an [B*T]-row GEMM with K = 80, N = 40 and
// the pooling kernel) this was 21 + 12 us of mostly latency.
__global__ __launch_bounds__(256) void attn_tail_fwd_kernel(
    int B, int T, int H, int N1, int N2, const float* __restrict__ a1, const float* __restrict__ W4,
    const float* __restrict__ b4, const float* __restrict__ w5, const float* __restrict__ b5,
    const int32_t* __restrict__ length, const float* __restrict__ ur, const float* __restrict__ ir,
    float* __restrict__ a2, float* __restrict__ score, float* __restrict__ head, int ldh, int off_u, int off_i) {
  extern __shared__ float sh[];
  const int L1 = N1 + 1, L2 = N2 + 1;
  float* a1s = sh;                 // [T][L1]
  float* w4s = a1s + T * L1;       // [N1][N2]
  float* a2s = w4s + N1 * N2;      // [T][L2]
  float* sc = a2s + T * L2;        // [T]
  __shared__ float red[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* a1b = a1 + (int64_t)b * T * N1;
  for (int i = tid; i < T * N1; i += 256) { const int t = i / N1; a1s[t * L1 + (i - t * N1)] = a1b[i]; }
  for (int i = tid; i < N1 * N2; i += 256) w4s[i] = W4[i];
  __syncthreads();
  float* a2b = a2 + (int64_t)b * T * N2;
  for (int i = tid; i < T * N2; i += 256) {
    const int t = i / N2, n = i - t * N2;
    float acc = 0.f;
#pragma unroll 8
    for (int k = 0; k < N1; ++k) acc = fmaf(a1s[t * L1 + k], w4s[k * N2 + n], acc);
    const float v = fmaxf(acc + b4[n], 0.f);
    a2s[t * L2 + n] = v;
    a2b[i] = v;
  }
  __syncthreads();
  const int len = length[b];
  for (int t = tid; t < T; t += 256) {
    float acc = 0.f;
    for (int n = 0; n < N2; ++n) acc = fmaf(a2s[t * L2 + n], w5[n], acc);
    sc[t] = t < len ? acc + b5[0] : -4294967295.0f;
  }
  __syncthreads();
  // softmax over T (max-subtracted), every thread walks the T scores in the same order
  float mx = -INFINITY;
  for (int t = 0; t < T; ++t) mx = fmaxf(mx, sc[t]);
  float den = 0.f;
  for (int t = 0; t < T; ++t) den += expf(sc[t] - mx);
  __syncthreads();
  for (int t = tid; t < T; t += 256) {
    const float v = expf(sc[t] - mx) / den;
    sc[t] = v;
    score[(int64_t)b * T + t] = v;
  }
  __syncthreads();
  (void)lane; (void)wave; (void)red;
  for (int j = tid; j < 2 * H; j += 256) {
    const bool us = j < H;
    const float* rep = (us ? ur : ir) + (int64_t)b * T * H + (us ? j : j - H);
    float s_ = 0.f;
    for (int t = 0; t < T; ++t) s_ = fmaf(rep[(int64_t)t * H], sc[t], s_);
    const int off = us ? off_u : off_i;
    if (off >= 0) head[(int64_t)b * ldh + off + (us ? j : j - H)] = s_;
  }
}

int score_launch_attn_tail_fwd(int B, int T, int H, int N1, int N2, const float* a1, const float* W4, const float* b4,
                               const float* w5, const float* b5, const int32_t* length, const float* ur, const float* ir,
                               float* a2, float* score, float* head, int ldh, int off_u, int off_i, hipStream_t s) {
  const size_t lds = (size_t)(T * (N1 + 1) + N1 * N2 + T * (N2 + 1) + T) * sizeof(float);
  if (lds > 60 * 1024) return SCORE_E_SHAPE;
  hipLaunchKernelGGL(attn_tail_fwd_kernel, dim3(B), dim3(256), lds, s, B, T, H, N1, N2, a1, W4, b4, w5, b5, length, ur, ir,
                     a2, score, head, ldh, off_u, off_i);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// backward of the pooling + masked softmax + fc3:  one block per sample, the T slices side by side
//   dscore_t = duf.ur_t + dif.ir_t ; ds_t = score_t (dscore_t - sum score*dscore) [t < len]
//   da2[t][n] = ds_t * w5[n] * [a2 > 0]
__global__ __launch_bounds__(256) void attn_pool_bwd_kernel(
    int B, int T, int H, int NA, int LPT, const float* __restrict__ a2, const float* __restrict__ w5,
    const int32_t* __restrict__ length, const float* __restrict__ ur, const float* __restrict__ ir,
    const float* __restrict__ score, const float* __restrict__ dhead, int ldh, int off_u, int off_i,
    float* __restrict__ ds, float* __restrict__ da2, int N1, const float* __restrict__ W4,
    const float* __restrict__ a1, float* __restrict__ da1) {
  extern __shared__ float sd[];   // [T] dscore  (+ [T][NA+1] da2 and [N1][NA+1] W4 when N1 > 0)
  const int b = blockIdx.x;
  const int len = length[b];
  const int gl = threadIdx.x % LPT;          // LPT lanes (a power of two <= 64) share one slice
  const float* du = off_u >= 0 ? dhead + (int64_t)b * ldh + off_u : nullptr;
  const float* di = off_i >= 0 ? dhead + (int64_t)b * ldh + off_i : nullptr;
  for (int t = threadIdx.x / LPT; t < T; t += 256 / LPT) {
    const int64_t bt = (int64_t)b * T + t;
    float part = 0.f;
    for (int j = gl; j < H; j += LPT) {
      if (du) part = fmaf(du[j], ur[bt * H + j], part);
      if (di) part = fmaf(di[j], ir[bt * H + j], part);
    }
    part = group_sum(part, LPT);
    if (gl == 0) sd[t] = part;
  }
  __syncthreads();
  float tot = 0.f;
  for (int t = 0; t < T; ++t) tot = fmaf(score[(int64_t)b * T + t], sd[t], tot);
  for (int i = threadIdx.x; i < T * NA; i += 256) {
    const int t = i / NA, n = i - t * NA;
    const int64_t bt = (int64_t)b * T + t;
    const float g = t < len ? score[bt] * (sd[t] - tot) : 0.f;
    const float dv = a2[bt * NA + n] > 0.f ? g * w5[n] : 0.f;
    da2[bt * NA + n] = dv;
    if (N1 > 0) sd[T + t * (NA + 1) + n] = dv;
    if (n == 0) ds[bt] = g;
  }
  if (N1 <= 0) return;
  // dense_4 backward in the same launch: da1[t][k] = [a1 > 0] sum_n da2[t][n] W4[k][n]  (the sample's da2 rows and the
  // layer's kernel in LDS; as its own [B*T]-row GEMM with K = 40 this was 14 us of mostly latency)
  float* d2s = sd + T;
  float* w4s = d2s + T * (NA + 1);
  for (int i = threadIdx.x; i < N1 * NA; i += 256) { const int k = i / NA; w4s[k * (NA + 1) + (i - k * NA)] = W4[i]; }
  __syncthreads();
  for (int i = threadIdx.x; i < T * N1; i += 256) {
    const int t = i / N1, k = i - t * N1;
    const int64_t e = ((int64_t)b * T + t) * N1 + k;
    float acc = 0.f;
#pragma unroll 8
    for (int n = 0; n < NA; ++n) acc = fmaf(d2s[t * (NA + 1) + n], w4s[k * (NA + 1) + n], acc);
    da1[e] = a1[e] > 0.f ? acc : 0.f;
  }
}

int score_launch_attn_pool_bwd(int B, int T, int H, int NA, const float* a2, const float* w5,
                               const int32_t* length, const float* ur, const float* ir, const float* score,
                               const float* dhead, int ldh, int off_u, int off_i, float* ds, float* da2,
                               hipStream_t s, int N1, const float* W4, const float* a1, float* da1) {
  // N1 > 0: also dense_4's backward, da1 = [a1 > 0] (da2 . W4^T) with W4 [N1][NA]; returns SCORE_E_SHAPE if that does
  // not fit LDS (the caller then runs it as a GEMM and calls again with N1 = 0)
  int LPT = 64;
  while (LPT > 1 && 256 / LPT < T) LPT >>= 1;      // as many slices side by side as the block holds
  size_t lds = (size_t)T * sizeof(float);
  if (N1 > 0) {
    lds += (size_t)(T + N1) * (NA + 1) * sizeof(float);
    if (lds > 60 * 1024) return SCORE_E_SHAPE;
  }
  hipLaunchKernelGGL(attn_pool_bwd_kernel, dim3(B), dim3(256), lds, s, B, T, H, NA, LPT, a2, w5,
                     length, ur, ir, score, dhead, ldh, off_u, off_i, ds, da2, N1, W4, a1, da1);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// backward of inp2 = [k, q*k] plus the pooled-state path into the GRU outputs; dq starts from the
// per-sample q-term gradient dqd = dzsum . (Wa + Wc)^T.  thread per (b, j); loops over t.
__global__ void attn_inp_bwd_kernel(int B, int T, int H, int NI, const float* __restrict__ dinp,
                                    const float* __restrict__ q, const float* __restrict__ ur,
                                    const float* __restrict__ ir, const float* __restrict__ info,
                                    const float* __restrict__ score, const float* __restrict__ dhead, int ldh,
                                    int off_u, int off_i, const float* __restrict__ dqd, float* __restrict__ dur,
                                    float* __restrict__ dir, float* __restrict__ dinfo, float* __restrict__ dq) {
  const int Dk = 2 * H + NI;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Dk) return;
  int b = i / Dk, j = i - b * Dk;
  float qq = q[(int64_t)b * Dk + j];
  float pooled = 0.f;
  if (j < H && off_u >= 0) pooled = dhead[(int64_t)b * ldh + off_u + j];
  if (j >= H && j < 2 * H && off_i >= 0) pooled = dhead[(int64_t)b * ldh + off_i + (j - H)];
  float dqa = 0.f;
  // source / destination segment of column j by pointer select (row stride H or NI): no branch around a load, so
  // the loads of several time slices are in flight together (they had been one dependent round trip per slice)
  const float* ksrc = j < H ? ur + j : (j < 2 * H ? ir + (j - H) : info + (j - 2 * H));
  float* kdst = j < H ? dur + j : (j < 2 * H ? dir + (j - H) : dinfo + (j - 2 * H));
  const int kst = j < 2 * H ? H : NI;
  const int64_t bt0 = (int64_t)b * T;
  constexpr int U = 6;
  for (int t0 = 0; t0 < T; t0 += U) {
    float kv[U], d1[U], d3[U], sc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t bt = bt0 + (t0 + u < T ? t0 + u : T - 1);
      kv[u] = ksrc[bt * kst];
      d1[u] = dinp[bt * 2 * Dk + j];
      d3[u] = dinp[bt * 2 * Dk + Dk + j];
      sc[u] = score[bt];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (t0 + u >= T) break;
      dqa = fmaf(d3[u], kv[u], dqa);
      kdst[(bt0 + t0 + u) * kst] = fmaf(d3[u], qq, d1[u]) + pooled * sc[u];    // (pooled = 0 for the atten_info columns)
    }
  }
  dq[(int64_t)b * Dk + j] = dqd ? dqa + dqd[(int64_t)b * Dk + j] : dqa;   // (null: the caller adds the q-term gradient itself)
}

int score_launch_attn_inp_bwd(int B, int T, int H, int NI, const float* dinp, const float* q, const float* ur,
                              const float* ir, const float* info, const float* score, const float* dhead, int ldh,
                              int off_u, int off_i, const float* dqd, float* dur, float* dir, float* dinfo, float* dq,
                              hipStream_t s) {
  int n = B * (2 * H + NI);
  hipLaunchKernelGGL(attn_inp_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, s, B, T, H, NI, dinp, q, ur, ir,
                     info, score, dhead, ldh, off_u, off_i, dqd, dur, dir, dinfo, dq);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// bn1 (inference-mode affine, score.py:69): y = x * (gamma * rs) + beta, rs = rsqrt(1 + 1e-3)
__global__ void bn_fwd_kernel(int B, int Dh, const float* __restrict__ x, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float rs, float* __restrict__ y) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * Dh) return;
  int j = (int)(i % Dh);
  y[i] = x[i] * (gamma[j] * rs) + beta[j];
}
// dx = dy * gamma*rs ; tmp = dy * x*rs  (dgamma = colsum(tmp), dbeta = colsum(dy))
__global__ void bn_bwd_kernel(int B, int Dh, const float* __restrict__ x, const float* __restrict__ gamma, float rs,
                              const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ tmp) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * Dh) return;
  int j = (int)(i % Dh);
  float d = dy[i];
  dx[i] = d * (gamma[j] * rs);
  tmp[i] = d * (x[i] * rs);
}

int score_launch_bn_fwd(int B, int Dh, const float* x, const float* gamma, const float* beta, float rs, float* y,
                        hipStream_t s) {
  int64_t n = (int64_t)B * Dh;
  hipLaunchKernelGGL(bn_fwd_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, B, Dh, x, gamma, beta, rs, y);
  SCORE_CHECK_LAUNCH();
  return 0;
}
int score_launch_bn_bwd(int B, int Dh, const float* x, const float* gamma, float rs, const float* dy, float* dx,
                        float* dgamma, float* dbeta, float* tmp, float* scratch, int64_t scratch_floats,
                        ColsumJobs* cq, hipStream_t s) {
  int64_t n = (int64_t)B * Dh;
  hipLaunchKernelGGL(bn_bwd_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, B, Dh, x, gamma, rs, dy, dx,
                     tmp);
  SCORE_CHECK_LAUNCH();
  if (cq) {
    SCORE_TRY(colsum_queue_add(cq, tmp, B, Dh, Dh, dgamma, 0));
    SCORE_TRY(colsum_queue_add(cq, dy, B, Dh, Dh, dbeta, 0));
  } else {
    SCORE_TRY(score_launch_colsum(tmp, B, Dh, Dh, dgamma, 0, scratch, scratch_floats, s));
    SCORE_TRY(score_launch_colsum(dy, B, Dh, Dh, dbeta, 0, scratch, scratch_floats, s));
  }
  return 0;
}

// fc3 + sigmoid + per-sample log-loss term and its gradient (score.py:74-81), thread per sample
//   loss_b = -y log(p+eps) - (1-y) log(1-p+eps) ; dlogit = dloss/dp * p(1-p) / B
__global__ void head_out_kernel(int B, int NF, const float* __restrict__ f2, const float* __restrict__ w3,
                                const float* __restrict__ b3, const int32_t* __restrict__ label,
                                float* __restrict__ logit, float* __restrict__ y, float* __restrict__ lossb,
                                float* __restrict__ dlogit, int Bglobal) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float acc = 0.f;
  for (int n = 0; n < NF; ++n) acc = fmaf(f2[(int64_t)b * NF + n], w3[n], acc);
  float z = acc + b3[0];
  float p = sigmoidf_(z);
  float lab = (float)label[b];
  const float eps = 1e-7f;
  logit[b] = z;
  y[b] = p;
  lossb[b] = -lab * logf(p + eps) - (1.0f - lab) * logf(1.0f - p + eps);
  float dp = (-lab / (p + eps) + (1.0f - lab) / (1.0f - p + eps)) / (float)Bglobal;
  dlogit[b] = dp * p * (1.0f - p);
}

#define L2_PARTS 256
// partial sums of squares of the regularised range (build_l2norm, score.py:91-94): 256 blocks, float4 loads
__device__ __forceinline__ void sumsq_stage1_body(int blk, int nblk, const float* __restrict__ x, int64_t n, float* __restrict__ part) {
  __shared__ float sh[256];
  const int64_t n4 = n >> 2;
  float s = 0.f;
  for (int64_t i = (int64_t)blk * 256 + threadIdx.x; i < n4; i += (int64_t)nblk * 256) {
    const float4 v = ld4(x + i * 4);
    s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
  }
  if (blk == 0 && threadIdx.x < (unsigned)(n - n4 * 4)) { const float v = x[n4 * 4 + threadIdx.x]; s = fmaf(v, v, s); }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blk] = sh[0];
}
__global__ __launch_bounds__(256) void sumsq_stage1(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
  sumsq_stage1_body(blockIdx.x, gridDim.x, x, n, part);
}
int score_launch_l2_partials(const float* wreg, int64_t n_reg, float* part /* L2_PARTS floats */, hipStream_t s) {
  hipLaunchKernelGGL(sumsq_stage1, dim3(L2_PARTS), dim3(256), 0, s, wreg, n_reg, part);
  SCORE_CHECK_LAUNCH();
  return 0;
}
// one block: loss[1] = sum_b lossb / Bglobal, loss[2] = 0.5 * sum(parts), loss[0] = loss[1] + lambda*loss[2]
// (fixed-order tree sums: reproducible)
__global__ __launch_bounds__(256) void loss_final_kernel(const float* __restrict__ lossb, int64_t B, float scale,
                                                         const float* __restrict__ part, float lambda,
                                                         float* __restrict__ loss,
                                                         const int32_t* __restrict__ id_status) {
  __shared__ float sh[256], sp[256];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < B; i += 256) s += lossb[i];
  sh[threadIdx.x] = s;
  sp[threadIdx.x] = part[threadIdx.x];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sh[threadIdx.x] += sh[threadIdx.x + o]; sp[threadIdx.x] += sp[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss[1] = sh[0] * scale;
    loss[2] = 0.5f * sp[0];
    loss[0] = loss[1] + lambda * loss[2];
    // an id outside the table (score_state_t.id_status; tf.nn.embedding_lookup raises there, score.py:51-66): the loss
    // of the step is poisoned so that whoever reads it learns of it without a second read-back; the bits ride in loss[3]
    const int32_t bad = id_status ? *id_status : 0;
    loss[3] = (float)bad;
    if (bad) loss[0] = loss[1] = __int_as_float(0x7fc00000);
  }
}

int score_launch_head_out(int B, int NF, const float* f2, const float* w3, const float* b3, const int32_t* label,
                          float* logit, float* y, float* lossb, float* dlogit, float* loss, float lambda,
                          const float* part /* L2_PARTS sums of squares from score_launch_l2_partials */, int Bglobal,
                          hipStream_t s, const int32_t* id_status) {
  // Bglobal = samples the mean is taken over (the local batch, or the global batch when data-parallel)
  hipLaunchKernelGGL(head_out_kernel, dim3((B + 63) / 64), dim3(64), 0, s, B, NF, f2, w3, b3, label, logit, y,
                     lossb, dlogit, Bglobal);
  SCORE_CHECK_LAUNCH();
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, lossb, (int64_t)B, 1.0f / (float)Bglobal, part, lambda,
                     loss, id_status);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// the loss reduction alone (the fused head kernel has already written lossb)
int score_launch_loss_final(int B, const float* lossb, float* loss, float lambda, const float* part, int Bglobal,
                            hipStream_t s, const int32_t* id_status) {
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, lossb, (int64_t)B, 1.0f / (float)Bglobal, part, lambda,
                     loss, id_status);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// dz[b][n] = [f[b][n] > 0] * dlogit[b] * w[n] / keep      (fc3 backward into relu+dropout of fc2)
__global__ void outer_relu_bwd_kernel(int B, int NF, const float* __restrict__ dlogit, const float* __restrict__ w,
                                      const float* __restrict__ f, float keep, float* __restrict__ dz) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * NF) return;
  int b = i / NF, n = i - b * NF;
  dz[i] = f[i] > 0.f ? dlogit[b] * w[n] / keep : 0.f;
}
int score_launch_outer_relu_bwd(int B, int NF, const float* dlogit, const float* w, const float* f, float keep,
                                float* dz, hipStream_t s) {
  hipLaunchKernelGGL(outer_relu_bwd_kernel, dim3((B * NF + 255) / 256), dim3(256), 0, s, B, NF, dlogit, w, f, keep,
                     dz);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// dst[r][c] = src[r][c]  with independent row strides
__global__ void copy2d_kernel(int64_t rows, int cols, const float* __restrict__ src, int lds_, float* __restrict__ dst,
                              int ldd) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  int64_t r = i / cols;
  int c = (int)(i - r * cols);
  dst[r * ldd + c] = src[r * lds_ + c];
}
int score_launch_copy2d(int64_t rows, int cols, const float* src, int lds_, float* dst, int ldd, hipStream_t s) {
  hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)cdiv64(rows * cols, 256)), dim3(256), 0, s, rows, cols, src,
                     lds_, dst, ldd);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// [Wx_gates | Wx_cand] and [b_gates | b_cand] of both GRUs side by side, so the hoisted input projection
// (and its two backward products) is ONE GEMM per side.  cat: [2][I+1][3H] (row I holds the bias).
struct WxcatArgs {
  const float* gk0; const float* ck0; const float* gb0; const float* cb0;
  const float* gk1; const float* ck1; const float* gb1; const float* cb1;
  int I0, I1, Imax, H; float* cat;
};
__device__ __forceinline__ void gru_wxcat_body(int blk, const WxcatArgs& a) {
  // side sd's block is [(Imax+1)][3H]: rows [0, I_sd) the x rows of both kernels, row I_sd the biases
  const int H = a.H;
  const int64_t per = (int64_t)(a.Imax + 1) * 3 * H;
  int64_t i = (int64_t)blk * 256 + threadIdx.x;
  if (i >= 2 * per) return;
  const int sd = i >= per;
  const int64_t l = i - sd * per;
  const int r = (int)(l / (3 * H)), j = (int)(l - (int64_t)r * 3 * H);
  const int I = sd ? a.I1 : a.I0;
  if (r > I) return;
  const float* gk = sd ? a.gk1 : a.gk0; const float* ck = sd ? a.ck1 : a.ck0;
  const float* gb = sd ? a.gb1 : a.gb0; const float* cb = sd ? a.cb1 : a.cb0;
  float v;
  if (r < I) v = j < 2 * H ? gk[(int64_t)r * 2 * H + j] : ck[(int64_t)r * H + (j - 2 * H)];
  else v = j < 2 * H ? gb[j] : cb[j - 2 * H];
  a.cat[i] = v;
}
__global__ __launch_bounds__(256) void gru_wxcat_kernel(const WxcatArgs a) { gru_wxcat_body(blockIdx.x, a); }

// The per-step transforms of the WEIGHTS -- the concatenated [Wx_gates | Wx_cand] copies, the folded first attention layer
// (optional) and the partial sums of squares of the regularised range -- in ONE launch (round 4: three launches before; the
// reference's own batch sizes are bound by the host's launch calls, ~6 us each).  Blocks [0, b_wx) copy, [b_wx, b_wx + b_fold)
// fold, the last L2_PARTS sum.
struct WeightPrepArgs {
  WxcatArgs wx; int b_wx;
  int Dk, NA; const float* W1; float* weff; float* wq; int copies; int64_t copy_stride; int b_fold;
  const float* wreg; int64_t n_reg; float* part;
};
__global__ __launch_bounds__(256) void weight_prep_kernel(const WeightPrepArgs a) {
  const int b = blockIdx.x;
  if (b < a.b_wx) gru_wxcat_body(b, a.wx);
  else if (b < a.b_wx + a.b_fold) attn_fold_w1_body(b - a.b_wx, a.Dk, a.NA, a.W1, a.weff, a.wq, a.copies, a.copy_stride);
  else sumsq_stage1_body(b - a.b_wx - a.b_fold, L2_PARTS, a.wreg, a.n_reg, a.part);
}
int score_launch_weight_prep(const float* gk0, const float* ck0, const float* gb0, const float* cb0, const float* gk1,
                             const float* ck1, const float* gb1, const float* cb1, int I0, int I1, int Imax, int H, float* cat,
                             int Dk, int NA, const float* W1, float* weff, float* wq, int copies, int64_t copy_stride,
                             const float* wreg, int64_t n_reg, float* part, hipStream_t s) {
  WeightPrepArgs a;
  a.wx = WxcatArgs{gk0, ck0, gb0, cb0, gk1, ck1, gb1, cb1, I0, I1, Imax, H, cat};
  a.b_wx = (int)cdiv64(2 * (int64_t)(Imax + 1) * 3 * H, 256);
  a.Dk = Dk; a.NA = NA; a.W1 = W1; a.weff = weff; a.wq = wq; a.copies = copies; a.copy_stride = copy_stride;
  a.b_fold = W1 ? (Dk * NA + 255) / 256 : 0;
  a.wreg = wreg; a.n_reg = n_reg; a.part = part;
  hipLaunchKernelGGL(weight_prep_kernel, dim3(a.b_wx + a.b_fold + L2_PARTS), dim3(256), 0, s, a);
  SCORE_CHECK_LAUNCH();
  return 0;
}
int score_launch_gru_wxcat(const float* gk0, const float* ck0, const float* gb0, const float* cb0, const float* gk1,
                           const float* ck1, const float* gb1, const float* cb1, int I0, int I1, int Imax, int H,
                           float* cat, hipStream_t s) {
  int64_t n = 2 * (int64_t)(Imax + 1) * 3 * H;
  hipLaunchKernelGGL(gru_wxcat_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s,
                     WxcatArgs{gk0, ck0, gb0, cb0, gk1, ck1, gb1, cb1, I0, I1, Imax, H, cat});
  SCORE_CHECK_LAUNCH();
  return 0;
}
// ------------------------------------------------------------------ ApplyAdam (score.py:96-99)
// TF training_ops: m += (g - m)(1-b1); v += (g*g - v)(1-b2); var -= m*alpha / (sqrt(v) + eps)
#define adam1 score_adam1     /* common.h */
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ g, int64_t n4, int64_t n, int64_t n_reg, float l2, float alpha,
                            float omb1, float omb2, float eps, const float* __restrict__ alpha_dev,
                            const int32_t* __restrict__ guard, int32_t* __restrict__ skipped) {
  // score_guard_t: a fed id outside the table -> the variables stay as they are (TF raises inside sess.run, score.py:51-66)
  if (guard && *guard) {
    if (skipped && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skipped, 1);
    return;
  }
  if (alpha_dev) alpha = *alpha_dev;          // score_step_scalars_t.adam_alpha (captured steps)
  score_adam_dense_body(p, m, v, g, n4, n, n_reg, l2, alpha, omb1, omb2, eps, (int)blockIdx.x, (int)gridDim.x);
}

// One group of D/4 lanes per table row; the state byte decides what the row costs (see score_hip.h).
// Two rows per group and trip: both state bytes, then both rows' streams are requested before anything is consumed
// (one row per trip left two dependent round trips per 48 bytes of a lane's traffic in flight).
#ifndef ADAM_NT
#define ADAM_NT 0
#endif
__device__ __forceinline__ float4 adam_ld(const float* p) {
#if ADAM_NT
  const score_v4f t = __builtin_nontemporal_load(reinterpret_cast<const score_v4f*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
#else
  return ld4(p);
#endif
}
__device__ __forceinline__ void adam_st(float* p, const float4& v) {
#if ADAM_NT
  score_v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<score_v4f*>(p));
#else
  st4(p, v);
#endif
}
__device__ __forceinline__ void adam_rows_body(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                               const float* __restrict__ g, int64_t n_rows, int D, int LPR,
                                               uint8_t* __restrict__ flags, float alpha, float omb1, float omb2, float eps,
                                               int blk, int nblk) {
  const int gpb = blockDim.x / LPR;
  const int ch4 = (threadIdx.x % LPR) * 4;
  const int64_t stride = (int64_t)nblk * gpb;
  int64_t row0 = (int64_t)blk * gpb + threadIdx.x / LPR;
  if (ch4 >= D) return;
  for (; row0 < n_rows; row0 += 2 * stride) {
    const int64_t row1 = row0 + stride;
    const bool has1 = row1 < n_rows;
    const uint8_t f0 = flags[row0];
    const uint8_t f1 = has1 ? flags[row1] : (uint8_t)0;
    const int64_t e0 = row0 * D + ch4, e1 = (has1 ? row1 : row0) * D + ch4;
    float4 p0, m0, v0, g0, p1, m1, v1, g1;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f0) { p0 = adam_ld(p + e0); m0 = adam_ld(m + e0); v0 = adam_ld(v + e0); }
    if (f1) { p1 = adam_ld(p + e1); m1 = adam_ld(m + e1); v1 = adam_ld(v + e1); }
    g0 = f0 == 2 ? adam_ld(g + e0) : z;
    g1 = f1 == 2 ? adam_ld(g + e1) : z;
    if (f0) {
      adam1(p0.x, m0.x, v0.x, g0.x, omb1, omb2, alpha, eps);
      adam1(p0.y, m0.y, v0.y, g0.y, omb1, omb2, alpha, eps);
      adam1(p0.z, m0.z, v0.z, g0.z, omb1, omb2, alpha, eps);
      adam1(p0.w, m0.w, v0.w, g0.w, omb1, omb2, alpha, eps);
      adam_st(p + e0, p0); adam_st(m + e0, m0); adam_st(v + e0, v0);
      // (every lane of the group read the byte above; a group is inside one wave, so the store below
      //  cannot overtake a sibling lane's load)
      if (f0 == 2 && ch4 == 0) flags[row0] = 1;
    }
    if (f1) {
      adam1(p1.x, m1.x, v1.x, g1.x, omb1, omb2, alpha, eps);
      adam1(p1.y, m1.y, v1.y, g1.y, omb1, omb2, alpha, eps);
      adam1(p1.z, m1.z, v1.z, g1.z, omb1, omb2, alpha, eps);
      adam1(p1.w, m1.w, v1.w, g1.w, omb1, omb2, alpha, eps);
      adam_st(p + e1, p1); adam_st(m + e1, m1); adam_st(v + e1, v1);
      if (f1 == 2 && ch4 == 0) flags[row1] = 1;
    }
  }
}
__global__ __launch_bounds__(256) void adam_rows_kernel(float* __restrict__ p, float* __restrict__ m,
                                                        float* __restrict__ v, const float* __restrict__ g,
                                                        int64_t n_rows, int D, int LPR, uint8_t* __restrict__ flags,
                                                        float alpha, float omb1, float omb2, float eps,
                                                        const float* __restrict__ alpha_dev,
                                                        const int32_t* __restrict__ guard, int32_t* __restrict__ skipped) {
  if (guard && *guard) {                       // score_guard_t (the state bytes stay: the caller clamps them)
    if (skipped && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skipped, 1);
    return;
  }
  if (alpha_dev) alpha = *alpha_dev;
  adam_rows_body(p, m, v, g, n_rows, D, LPR, flags, alpha, omb1, omb2, eps, (int)blockIdx.x, (int)gridDim.x);
}
// score_adam_rows on the table AND score_adam on the flat dense variables in ONE launch (round 5: the per-step sweep of a small
// table -- cfg-2 -- ended in two dependent launches of 11 and 4 us with a launch boundary between them): workgroups [0, nb_r) the
// rows, the rest the dense variables; disjoint memory, the same arithmetic per element as the two calls.  A set guard word:
// neither half applies anything, the dense half counts the suppressed step (score_guard_t: ONE counted call per step).
struct AdamDenseHalf { float* p; float* m; float* v; const float* g; int64_t n4, n, n_reg; float l2; };
__global__ __launch_bounds__(256) void adam_rows_dense_kernel(float* __restrict__ p, float* __restrict__ m,
                                                              float* __restrict__ v, const float* __restrict__ g,
                                                              int64_t n_rows, int D, int LPR, uint8_t* __restrict__ flags,
                                                              float alpha, float omb1, float omb2, float eps, const AdamDenseHalf d,
                                                              int nb_r, int nb_d, const int32_t* __restrict__ guard,
                                                              int32_t* __restrict__ skipped) {
  if (guard && *guard) {
    if (skipped && (int)blockIdx.x == nb_r && threadIdx.x == 0) atomicAdd(skipped, 1);
    return;
  }
  if ((int)blockIdx.x < nb_r) {
    adam_rows_body(p, m, v, g, n_rows, D, LPR, flags, alpha, omb1, omb2, eps, (int)blockIdx.x, nb_r);
    return;
  }
  score_adam_dense_body(d.p, d.m, d.v, d.g, d.n4, d.n, d.n_reg, d.l2, alpha, omb1, omb2, eps, (int)blockIdx.x - nb_r, nb_d);
}

static int adam_rows_impl(float* p, float* m, float* v, const float* g, int64_t n_rows, int32_t D,
                          uint8_t* row_flags, float alpha, const float* alpha_dev, float beta1, float beta2, float eps,
                          const score_guard_t* guard, void* stream) {
  if (!p || !m || !v || !g || !row_flags || n_rows <= 0 || D <= 0) return SCORE_E_BADARG;
  if ((D & 3) || D > 256) return SCORE_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
       reinterpret_cast<uintptr_t>(g)) & 15)
    return SCORE_E_SHAPE;
  int LPR = 1;
  while (LPR < D / 4) LPR <<= 1;
  const int gpb = 256 / LPR;
  int64_t want = cdiv64(n_rows, gpb);
  int blocks = (int)(want < 16384 ? want : 16384);
  hipLaunchKernelGGL(adam_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, m, v, g, n_rows, D, LPR,
                     row_flags, alpha, 1.0f - beta1, 1.0f - beta2, eps, alpha_dev, guard ? guard->id_status : nullptr,
                     guard ? guard->skipped : nullptr);
  SCORE_CHECK_LAUNCH();
  return 0;
}
extern "C" int score_adam_rows(float* p, float* m, float* v, const float* g, int64_t n_rows, int32_t D,
                               uint8_t* row_flags, float alpha, float beta1, float beta2, float eps,
                               const score_guard_t* guard, void* stream) {
  return adam_rows_impl(p, m, v, g, n_rows, D, row_flags, alpha, nullptr, beta1, beta2, eps, guard, stream);
}
extern "C" int score_adam_rows_and_dense(float* p, float* m, float* v, const float* g, int64_t n_rows, int32_t D,
                                         uint8_t* row_flags, float* wp, float* wm, float* wv, const float* wg, int64_t n,
                                         int64_t n_reg, float l2, float alpha, float beta1, float beta2, float eps,
                                         const score_guard_t* guard, void* stream) {
  if (!p || !m || !v || !g || !row_flags || n_rows <= 0 || D <= 0 || !wp || !wm || !wv || !wg || n <= 0) return SCORE_E_BADARG;
  if ((D & 3) || D > 256) return SCORE_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
       reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(wp) | reinterpret_cast<uintptr_t>(wm) |
       reinterpret_cast<uintptr_t>(wv) | reinterpret_cast<uintptr_t>(wg)) & 15)
    return SCORE_E_SHAPE;
  int LPR = 1;
  while (LPR < D / 4) LPR <<= 1;
  const int gpb = 256 / LPR;
  const int64_t want_r = cdiv64(n_rows, gpb);
  const int nb_r = (int)(want_r < 16384 ? want_r : 16384);
  AdamDenseHalf d;
  d.p = wp; d.m = wm; d.v = wv; d.g = wg; d.n4 = n / 4; d.n = n; d.n_reg = n_reg; d.l2 = l2;
  const int64_t want_d = cdiv64(d.n4 > 0 ? d.n4 : 1, 256);
  const int nb_d = (int)(want_d < 8192 ? want_d : 8192);
  hipLaunchKernelGGL(adam_rows_dense_kernel, dim3(nb_r + nb_d), dim3(256), 0, (hipStream_t)stream, p, m, v, g, n_rows, D, LPR,
                     row_flags, alpha, 1.0f - beta1, 1.0f - beta2, eps, d, nb_r, nb_d, guard ? guard->id_status : nullptr,
                     guard ? guard->skipped : nullptr);
  SCORE_CHECK_LAUNCH();
  return 0;
}
extern "C" int score_adam_rows_dev(float* p, float* m, float* v, const float* g, int64_t n_rows, int32_t D,
                                   uint8_t* row_flags, const score_step_scalars_t* sc, float beta1, float beta2,
                                   float eps, const score_guard_t* guard, void* stream) {
  if (!sc) return SCORE_E_BADARG;
  return adam_rows_impl(p, m, v, g, n_rows, D, row_flags, 0.f, &sc->adam_alpha, beta1, beta2, eps, guard, stream);
}

static int adam_impl(float* p, float* m, float* v, const float* g, int64_t n, int64_t n_reg, float l2,
                     float alpha, const float* alpha_dev, float beta1, float beta2, float eps, const score_guard_t* guard,
                     void* stream) {
  if (!p || !m || !v || !g || n <= 0) return SCORE_E_BADARG;
  if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
       reinterpret_cast<uintptr_t>(g)) & 15)
    return SCORE_E_SHAPE;
  int64_t n4 = n / 4;
  int64_t want = cdiv64(n4 > 0 ? n4 : 1, 256);
  int blocks = (int)(want < 8192 ? want : 8192);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, m, v, g, n4, n, n_reg, l2,
                     alpha, 1.0f - beta1, 1.0f - beta2, eps, alpha_dev, guard ? guard->id_status : nullptr,
                     guard ? guard->skipped : nullptr);
  SCORE_CHECK_LAUNCH();
  return 0;
}
extern "C" int score_adam(float* p, float* m, float* v, const float* g, int64_t n, int64_t n_reg, float l2,
                          float alpha, float beta1, float beta2, float eps, const score_guard_t* guard, void* stream) {
  return adam_impl(p, m, v, g, n, n_reg, l2, alpha, nullptr, beta1, beta2, eps, guard, stream);
}
extern "C" int score_adam_dev(float* p, float* m, float* v, const float* g, int64_t n, int64_t n_reg, float l2,
                              const score_step_scalars_t* sc, float beta1, float beta2, float eps,
                              const score_guard_t* guard, void* stream) {
  if (!sc) return SCORE_E_BADARG;
  return adam_impl(p, m, v, g, n, n_reg, l2, 0.f, &sc->adam_alpha, beta1, beta2, eps, guard, stream);
}

// ---------------------------------------------------------------- stream copy (measurement helper, score_hip.h)
// A block owns one contiguous chunk; four 16-byte loads per lane in flight before the first store.  The shape that came
// out on top of tools/copy_probe.py on this pool (1 GiB each way: 5.3 - 5.6 TB/s; grid-stride forms 4.4 - 5.4, the
// runtime's own device-to-device copy 4.6 - 4.8).
__global__ __launch_bounds__(256) void stream_copy_kernel(float4* __restrict__ dst, const float4* __restrict__ src, int64_t n4,
                                                          int64_t chunk) {
  const int64_t lo = (int64_t)blockIdx.x * chunk;
  const int64_t hi = lo + chunk < n4 ? lo + chunk : n4;
  int64_t i = lo + threadIdx.x;
  for (; i + 3 * 256 < hi; i += 4 * 256) {
    const float4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
    dst[i] = a; dst[i + 256] = b; dst[i + 512] = c; dst[i + 768] = d;
  }
  for (; i < hi; i += 256) dst[i] = src[i];
}
extern "C" int score_stream_copy(float* dst, const float* src, int64_t n_floats, void* stream) {
  if (!dst || !src || n_floats <= 0) return SCORE_E_BADARG;
  if ((n_floats & 3) || ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15)) return SCORE_E_SHAPE;
  const int64_t n4 = n_floats / 4;
  int64_t blocks = cdiv64(n4, 1024);
  if (blocks > 32768) blocks = 32768;
  const int64_t chunk = cdiv64(n4, blocks);
  hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<float4*>(dst), reinterpret_cast<const float4*>(src), n4, chunk);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------- table initialiser (score.py:44)
// Truncated normal(0, 1) on [-2, 2] by inverse CDF: u uniform on (Phi(-2), Phi(2)), x = sqrt(2) * erfinv(2u - 1).
// The uniform is a counter hash of (seed, global element index): the value of an element does not depend on
// which shard holds it or on the launch geometry.
__global__ __launch_bounds__(256) void table_init_kernel(float* __restrict__ table, int64_t n_local, int D,
                                                         int64_t stride, int64_t first, int64_t n_global,
                                                         uint64_t seed) {
  const int64_t n = n_local * D;
  const float lo = 0.02275013194817921f, span = 0.9544997361036416f;     // Phi(-2), Phi(2) - Phi(-2)
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t lr = e / D, col = e - lr * D;
    const int64_t gr = lr * stride + first;
    float x = 0.f;
    if (gr != 0 && gr < n_global) {
      // two 24-bit hashes -> a 48-bit uniform, so the tails are resolved finer than fp32's 2^-24 grid
      const uint64_t idx = (uint64_t)gr * (uint64_t)D + (uint64_t)col;
      const double u = ((double)hash_uniform(seed, 2 * idx) + (double)hash_uniform(seed ^ 0xD1B54A32D192ED03ull, 2 * idx + 1) *
                        (1.0 / 16777216.0)) + (0.5 / 281474976710656.0);
      x = 1.4142135623730951f * erfinvf((float)(2.0 * ((double)lo + (double)span * u) - 1.0));
      x = fminf(fmaxf(x, -2.f), 2.f);
    }
    table[e] = x;
  }
}

extern "C" int score_table_init(float* table, int64_t n_local_rows, int32_t D, int64_t row_stride, int64_t row_first,
                                int64_t n_global_rows, uint64_t seed, void* stream) {
  if (!table || n_local_rows <= 0 || D <= 0 || row_stride <= 0 || row_first < 0 || row_first >= row_stride ||
      n_global_rows <= 0)
    return SCORE_E_BADARG;
  const int64_t want = cdiv64(n_local_rows * D, 256 * 8);
  const int blocks = (int)(want < 65536 ? (want > 0 ? want : 1) : 65536);
  hipLaunchKernelGGL(table_init_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, n_local_rows, (int)D,
                     row_stride, row_first, n_global_rows, seed);
  SCORE_CHECK_LAUNCH();
  return 0;
}
