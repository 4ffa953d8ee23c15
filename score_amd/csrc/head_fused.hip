// build_fc_net forward in ONE launch (score.py:68-81): bn1 (inference-mode affine) -> fc1 200 relu dropout -> fc2 80
// relu dropout -> fc3 -> sigmoid -> per-sample log-loss term and its gradient.
//
// As separate launches the head was five kernels of M = B = 1024 rows whose time is the latency of their K loops
// (0.062 ms of a 1.57 ms step).  Here a workgroup of 8 waves owns 16 samples: their bn1 output sits in LDS, the three
// layers run back to back on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation) with the weights
// streamed from L2, every intermediate a later pass needs (bn1 output, f1, f2, logit, y_pred, the loss term and
// dlogit) is written once on the way.  K is dealt to the four lane quarters in contiguous runs, so a lane reads its
// A operands of four consecutive MFMA steps with one ds_read_b128; the B operands (one weight per lane and step,
// 64-B runs per 16 lanes) are fetched a 16-step chunk ahead.
#include "common.h"
#include "kernels.h"

typedef float hf_f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int HF_ROWS = 16;     // samples per workgroup
constexpr int HF_NW = 8;        // waves per workgroup
constexpr int HF_CH = 16;       // MFMA steps per prefetched chunk of B operands

// acc[t] (16 x 16, rows lq*4+r, column lc) += xs[16][Kp] . W[K][ldw] columns [n0[t], n0[t]+16) for NT tiles at once.
// xs: LDS, row stride LD, zero beyond K up to Kp (Kp % 16 == 0, so KQ = Kp/4 is a multiple of 4).
// (The weight operand read from a transposed copy, one 16-B load per lane and four steps, measured slower -- 0.056 vs
// 0.051 ms for the stage: 16 lanes then touch 16 different rows per load instead of one 64-B run.)
template <int NT>
__device__ __forceinline__ void hf_tiles(hf_f32x4 (&acc)[NT], const float* __restrict__ xs, int LD, int Kp, int K,
                                         const float* __restrict__ W, int ldw, const int (&n0)[NT], int N, int lc, int lq) {
  const int KQ = Kp >> 2;
  const int kbase = lq * KQ;
  int col[NT];
  bool cok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    cok[t] = n0[t] >= 0 && n0[t] + lc < N;
    col[t] = cok[t] ? n0[t] + lc : 0;
  }
  // B operands of two chunks in registers: chunk c+1 is requested before chunk c's MFMAs are issued
  float b0[NT][HF_CH], b1[NT][HF_CH];
  auto fetch = [&](float (&bb)[NT][HF_CH], int s0) {
#pragma unroll
    for (int i = 0; i < HF_CH; ++i) {
      const int k = kbase + s0 + i;
      const int kc = k < K ? k : K - 1;                 // clamped, unconditional loads; masked below
#pragma unroll
      for (int t = 0; t < NT; ++t) bb[t][i] = W[(int64_t)kc * ldw + col[t]];
    }
  };
  const float* xrow = xs + lc * LD + kbase;
  auto compute = [&](const float (&bb)[NT][HF_CH], int s0) {
    float4 av[HF_CH / 4];
#pragma unroll
    for (int q = 0; q < HF_CH / 4; ++q) {
      const int s = s0 + 4 * q;
      av[q] = s < KQ ? *reinterpret_cast<const float4*>(xrow + s) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < HF_CH; ++i) {
      const int k = kbase + s0 + i;
      const bool kok = (s0 + i < KQ) && k < K;
      const float a = (i & 3) == 0 ? av[i >> 2].x : (i & 3) == 1 ? av[i >> 2].y : (i & 3) == 2 ? av[i >> 2].z : av[i >> 2].w;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float b = (kok && cok[t]) ? bb[t][i] : 0.f;
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kok ? a : 0.f, b, acc[t], 0, 0, 0);
      }
    }
  };
  fetch(b0, 0);
  for (int s0 = 0; s0 < KQ; s0 += 2 * HF_CH) {
    fetch(b1, s0 + HF_CH);          // (addresses past the quarter are clamped and their products masked)
    compute(b0, s0);
    fetch(b0, s0 + 2 * HF_CH);
    if (s0 + HF_CH < KQ) compute(b1, s0 + HF_CH);
  }
}

struct HeadFwdArgs {
  int B, Dh, N1, N2, Bglobal;
  const float* x; const float* gamma; const float* beta; float rs;
  const float* W1; const float* b1; const float* W2; const float* b2; const float* W3; const float* b3;
  float keep; int drop; const uint8_t* mask0; const uint8_t* mask1; uint64_t seed0, seed1;
  const uint64_t* seed_dev;       // score_step_scalars_t.drop_seed (captured steps): overrides seed0 / seed1
  const int32_t* label;
  float* bn; float* f1; float* f2; float* logit; float* y; float* lossb; float* dlogit;
  float* dz2;                     // [B, N2] fc3's backward into relu+dropout of fc2 (what the backward pass starts from)
  // phase 0: the whole head in one launch (small batches: launch-bound).  Batches of many 16-row tiles run it as two:
  // phase 1 = bn1 + fc1 with the column tiles of fc1 dealt to gridDim.y workgroups per row tile (a row tile's fc1 is
  // 563 KB of weights streamed by ONE workgroup otherwise, on B/16 of the chip's CUs), phase 2 = fc2, fc3, loss, dz2
  // from the saved fc1 output.  Same arithmetic per element either way.
  int phase;
};

__device__ __forceinline__ float hf_act(float v, float bias, int drop, float keep, const uint8_t* mask, uint64_t seed,
                                        int row, int col, int N) {
  v = fmaxf(v + bias, 0.f);                             // dense(activation=relu)
  if (drop) {                                           // tf.nn.dropout: x / keep * Bernoulli(keep)  (same element
    const uint64_t e = (uint64_t)row * (uint64_t)N + (uint64_t)col;      // numbering as the GEMM epilogue's)
    const bool on = mask ? (mask[e] != 0) : (hash_uniform(seed, e) < keep);
    v = on ? v / keep : 0.f;
  }
  return v;
}

__global__ __launch_bounds__(64 * HF_NW) void head_fwd_fused_kernel(const HeadFwdArgs a) {
  extern __shared__ float sm[];
  const uint64_t seed0 = a.seed_dev ? *a.seed_dev : a.seed0;
  const uint64_t seed1 = a.seed_dev ? (seed0 ^ 0x5DEECE66Dull) : a.seed1;
  const int Dh = a.Dh, N1 = a.N1, N2 = a.N2;
  const int Kp0 = (Dh + 15) & ~15, LD0 = Kp0 + 4;
  const int Kp1 = (N1 + 15) & ~15, LD1 = Kp1 + 4;
  const int Kp2 = (N2 + 15) & ~15, LD2 = Kp2 + 4;
  float* xs = sm;                          // [16][LD0]  bn1 output
  float* f1s = xs + HF_ROWS * LD0;         // [16][LD1]
  float* f2s = f1s + HF_ROWS * LD1;        // [16][LD2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, lq = lane >> 4;
  const int b0 = blockIdx.x * HF_ROWS;

  const int phase = a.phase;
  const bool store_bn = blockIdx.y == 0;
  // bn1: y = x * gamma * rs + beta  (moving mean 0 / variance 1, never updated: score.py:69 runs it in inference mode).
  // float4 per thread and trip, four trips' loads in flight together (clamped addresses, no branch around a load)
  for (int e = tid; e < HF_ROWS * LD0; e += 64 * HF_NW) xs[e] = 0.f;
  __syncthreads();
  if (phase == 2) {
    // (phase 2 starts from the saved fc1 output)
  } else if ((Dh & 3) == 0) {
    const int n4 = Dh >> 2, total = HF_ROWS * n4;
    for (int e0 = tid; e0 < total; e0 += 4 * 64 * HF_NW) {
      float4 xv[4], gv[4], bv[4];
      int ii[4], jj[4];
      bool okv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * 64 * HF_NW;
        const int ec = e < total ? e : 0;
        ii[u] = ec / n4; jj[u] = (ec - ii[u] * n4) * 4;
        okv[u] = e < total && b0 + ii[u] < a.B;
        const int row = b0 + ii[u] < a.B ? b0 + ii[u] : a.B - 1;
        xv[u] = ld4(a.x + (int64_t)row * Dh + jj[u]);
        gv[u] = ld4(a.gamma + jj[u]);
        bv[u] = ld4(a.beta + jj[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!okv[u]) continue;
        float4 v;
        v.x = xv[u].x * (gv[u].x * a.rs) + bv[u].x; v.y = xv[u].y * (gv[u].y * a.rs) + bv[u].y;
        v.z = xv[u].z * (gv[u].z * a.rs) + bv[u].z; v.w = xv[u].w * (gv[u].w * a.rs) + bv[u].w;
        if (store_bn) st4(a.bn + (int64_t)(b0 + ii[u]) * Dh + jj[u], v);
        *reinterpret_cast<float4*>(xs + ii[u] * LD0 + jj[u]) = v;
      }
    }
  } else {
    for (int e = tid; e < HF_ROWS * Dh; e += 64 * HF_NW) {
      const int i = e / Dh, j = e - i * Dh;
      if (b0 + i < a.B) {
        const float v = a.x[(int64_t)(b0 + i) * Dh + j] * (a.gamma[j] * a.rs) + a.beta[j];
        if (store_bn) a.bn[(int64_t)(b0 + i) * Dh + j] = v;
        xs[i * LD0 + j] = v;
      }
    }
  }
  for (int e = tid; e < HF_ROWS * (LD1 + LD2); e += 64 * HF_NW) f1s[e] = 0.f;      // zero padding of the next layers' K
  __syncthreads();

  const int nt1 = (N1 + 15) >> 4;
  if (phase == 1) {       // this workgroup's share of fc1's column tiles, one per wave and pass; straight to global memory
    const int tpg = (nt1 + (int)gridDim.y - 1) / (int)gridDim.y;
    const int tend = min(nt1, ((int)blockIdx.y + 1) * tpg);
    for (int tb = (int)blockIdx.y * tpg; tb < tend; tb += HF_NW) {
      hf_f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
      const int t0 = tb + wave;
      const int n0[1] = {t0 < tend ? t0 * 16 : -1};
      if (n0[0] >= 0) hf_tiles<1>(acc, xs, LD0, Kp0, Dh, a.W1, N1, n0, N1, lc, lq);
      const int col = n0[0] + lc;
      const float bias = a.b1[min(max(col, 0), N1 - 1)];
      if (n0[0] >= 0 && col < N1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = b0 + lq * 4 + r;
          const float v = hf_act(acc[0][r], bias, a.drop, a.keep, a.mask0, seed0, row, col, N1);
          if (row < a.B) a.f1[(int64_t)row * N1 + col] = v;
        }
      }
    }
    return;
  }
  if (phase == 2) {       // the saved fc1 output of this row tile -> LDS (its padding is zero already)
    for (int e = tid; e < HF_ROWS * N1; e += 64 * HF_NW) {
      const int i = e / N1, j = e - i * N1;
      const int row = min(b0 + i, a.B - 1);
      f1s[i * LD1 + j] = a.f1[(int64_t)row * N1 + j];
    }
  }
  // fc1: tiles of 16 columns, two per wave and pass
  for (int tb = 0; phase == 0 && tb < nt1; tb += 2 * HF_NW) {
    hf_f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int t0 = tb + wave, t1 = tb + HF_NW + wave;
    const int n0[2] = {t0 < nt1 ? t0 * 16 : -1, t1 < nt1 ? t1 * 16 : -1};
    if (n0[0] >= 0) hf_tiles<2>(acc, xs, LD0, Kp0, Dh, a.W1, N1, n0, N1, lc, lq);
    float bias1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) bias1[t] = a.b1[min(max(n0[t] + lc, 0), N1 - 1)];     // clamped, unconditional (no load behind a branch)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int col = n0[t] + lc;
      if (n0[t] < 0 || col >= N1) continue;
      const float bias = bias1[t];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = lq * 4 + r, row = b0 + i;
        const float v = hf_act(acc[t][r], bias, a.drop, a.keep, a.mask0, seed0, row, col, N1);
        f1s[i * LD1 + col] = v;
        if (row < a.B) a.f1[(int64_t)row * N1 + col] = v;
      }
    }
  }
  __syncthreads();

  // fc2
  const int nt2 = (N2 + 15) >> 4;
  for (int tb = 0; tb < nt2; tb += HF_NW) {
    hf_f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
    const int t0 = tb + wave;
    const int n0[1] = {t0 < nt2 ? t0 * 16 : -1};
    if (n0[0] >= 0) hf_tiles<1>(acc, f1s, LD1, Kp1, N1, a.W2, N2, n0, N2, lc, lq);
    const int col = n0[0] + lc;
    const float bias2 = a.b2[min(max(col, 0), N2 - 1)];
    if (n0[0] >= 0 && col < N2) {
      const float bias = bias2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = lq * 4 + r, row = b0 + i;
        const float v = hf_act(acc[0][r], bias, a.drop, a.keep, a.mask1, seed1, row, col, N2);
        f2s[i * LD2 + col] = v;
        if (row < a.B) a.f2[(int64_t)row * N2 + col] = v;
      }
    }
  }
  __syncthreads();

  // fc3 + sigmoid + log-loss term and its gradient: four lanes per sample, fixed-order partial sums
  if (wave == 0) {
    const int i = lane >> 2, part = lane & 3;
    float s = 0.f;
    for (int n = part; n < N2; n += 4) s = fmaf(f2s[i * LD2 + n], a.W3[n], s);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    const int row = b0 + i;
    if (part == 0 && row < a.B) {
      const float z = s + a.b3[0];
      const float p = sigmoidf_(z);
      const float lab = (float)a.label[row];
      const float eps = 1e-7f;
      a.logit[row] = z;
      a.y[row] = p;
      a.lossb[row] = -lab * logf(p + eps) - (1.0f - lab) * logf(1.0f - p + eps);
      const float dp = (-lab / (p + eps) + (1.0f - lab) / (1.0f - p + eps)) / (float)a.Bglobal;
      const float dl = dp * p * (1.0f - p);
      a.dlogit[row] = dl;
      xs[i] = dl;                       // (bn1's tile is dead by now)
    }
  }
  __syncthreads();
  // dz2[b][n] = [f2 > 0] * dlogit[b] * w3[n] / keep: the first thing the backward pass needs, and everything it is made
  // of is here (it was a launch of its own at the head of score_backward).  (The loss reduction did NOT move in here:
  // done by the last workgroup to arrive, it cost this kernel 6.6 us -- more than the one-block launch behind it.)
  if (a.dz2) {
    for (int e = tid; e < HF_ROWS * N2; e += 64 * HF_NW) {
      const int i = e / N2, n = e - i * N2, row = b0 + i;
      const float q = xs[i] * a.W3[n] / a.keep;                   // (computed for every element: no load behind the branch)
      if (row < a.B) a.dz2[(int64_t)row * N2 + n] = f2s[i * LD2 + n] > 0.f ? q : 0.f;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Temporal attention forward in ONE launch (score.py:169-186, 210-215), a workgroup per group of S samples:
//   inp rows [k, q o k] (k = [user state | item state | atten_info] of a slice, q the sample's projected query) -> LDS
//   (and to global memory: the backward pass reads them) -> dense_3, folded (a1 = relu(inp . Weff + qz[sample])) on
//   v_mfma_f32_16x16x4_f32 with the weights streamed from L2 (hf_tiles) -> dense_4 -> dense_5 -> mask -> softmax over
//   the slices -> pooled states into the head's input.
// As four launches (build [B*T, 2Dk] in global memory, a bf16x3 GEMM with N = 80 on 128-wide tiles plus its split-K
// reduce, the tail kernel) this was 0.018 + 0.046 + 0.019 + 0.023 ms at cfg-3; the [B*T, 2Dk] matrix went out to HBM
// and came straight back.  Here the rows of S samples form M-tiles of 16 (rows of different samples share a tile: the
// fold puts the per-sample q into the A operand, the weight is common); waves 0..4 own the five 16-column tiles of
// dense_3, waves 5..7 build the next M-tile meanwhile (two tile buffers).
constexpr int AF_NW = 8;

struct AttnFwdArgs {
  int B, T, H, NI, N1, N2, S;
  const float* q; const float* ur; const float* ir; const float* info;
  const float* Weff; const float* qz;
  const float* W4; const float* b4; const float* w5; const float* b5; const int32_t* length;
  float* inp; float* a1; float* a2; float* score; float* head; int ldh, off_u, off_i;
  int wcopies; int64_t wstride;   // replicas of Weff (head.hip: attn_fold_w1_kernel)
};

__device__ __forceinline__ void af_build(const AttnFwdArgs& a, float* __restrict__ xs, int LD, int Kp, int row0,
                                         int row_end, int t0, int nthr) {
  const int Dk = 2 * a.H + a.NI, Dk4 = Dk >> 2, K2 = 2 * Dk;
  const int total = 16 * Dk4;
  constexpr int U = 4;                 // items per trip: their loads go out together (one dependent round trip per item
                                       // made the three building waves the slowest part of the kernel)
  for (int e0 = t0; e0 < total; e0 += U * nthr) {
    float4 k[U], qq[U];
    int ii[U], jj[U], gg[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * nthr;
      const int ec = e < total ? e : 0;
      ii[u] = ec / Dk4; jj[u] = (ec - ii[u] * Dk4) * 4;
      gg[u] = row0 + ii[u];
      ok[u] = e < total && gg[u] < row_end;
      const int g = gg[u] < row_end ? gg[u] : row_end - 1;           // clamped, unconditional loads
      const int j = jj[u];
      const float* src = j < a.H ? a.ur + (int64_t)g * a.H + j
                                 : (j < 2 * a.H ? a.ir + (int64_t)g * a.H + (j - a.H) : a.info + (int64_t)g * a.NI + (j - 2 * a.H));
      k[u] = ld4(src);
      qq[u] = ld4(a.q + (int64_t)(g / a.T) * Dk + j);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (e0 + u * nthr >= total) continue;
      float4 kk = k[u], qk = make_float4(qq[u].x * kk.x, qq[u].y * kk.y, qq[u].z * kk.z, qq[u].w * kk.w);
      if (ok[u]) {
#ifndef AFP_NOINP
        st4(a.inp + (int64_t)gg[u] * K2 + jj[u], kk);
        st4(a.inp + (int64_t)gg[u] * K2 + Dk + jj[u], qk);
#endif
      } else {
        kk = make_float4(0.f, 0.f, 0.f, 0.f); qk = kk;
      }
      *reinterpret_cast<float4*>(xs + ii[u] * LD + jj[u]) = kk;
      *reinterpret_cast<float4*>(xs + ii[u] * LD + Dk + jj[u]) = qk;
    }
  }
  for (int e = t0; e < 16 * (Kp - K2); e += nthr) {      // zero padding of K up to a multiple of 16
    const int i = e / (Kp - K2), j = e - i * (Kp - K2);
    xs[i * LD + K2 + j] = 0.f;
  }
}

// KQ = Kp / 4: k-steps per lane quarter.  A wave of the dense_3 phase keeps its tile's B operands -- one weight per
// lane and k-step, KQ registers -- for the whole launch (streamed from L2 per M-tile, 10 chunks of 16 dependent-latency
// loads each, that phase alone was ~50 us: the kernel measured 102 us against 78-95 for the launches it replaces).
template <int KQ>
__global__ __launch_bounds__(64 * AF_NW) void attn_fwd_fused_kernel(const AttnFwdArgs a) {
  extern __shared__ float sm[];
  const int T = a.T, H = a.H, N1 = a.N1, N2 = a.N2;
  const int Dk = 2 * H + a.NI, K2 = 2 * Dk, Kp = 4 * KQ, LD = Kp + 4;
  const int L1 = N1 + 1, L2 = N2 + 1;
  float* xs0 = sm;                              // [16][LD] two M-tile buffers
  float* xs1 = xs0 + 16 * LD;
  float* a1s = xs1 + 16 * LD;                   // [S*T][L1]
  float* w4s = a1s + a.S * T * L1;              // [N1][N2]
  float* a2s = w4s + N1 * N2;                   // [S][T][L2]
  float* scs = a2s + a.S * T * L2;              // [S][T]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, lq = lane >> 4;
  const int b0 = blockIdx.x * a.S;
  const int ns = min(a.S, a.B - b0);
  const int row0 = b0 * T, row_end = row0 + ns * T;      // (B*T < 2^31: checked by the launcher)
  const int ntile = (ns * T + 15) >> 4;
  const int ntn = (N1 + 15) >> 4;               // <= 5 (checked by the launcher)

  float breg[KQ];
  if (wave < ntn) {
    const int col = wave * 16 + lc;
    // (workgroups b, b+8, ... share an XCD and its L2: they take different replicas)
    const float* Wsrc = a.Weff + (int64_t)((blockIdx.x >> 3) % a.wcopies) * a.wstride;
#pragma unroll
    for (int s_ = 0; s_ < KQ; ++s_) {
      const int k = lq * KQ + s_;
#ifdef AFP_NOPRELOAD
      const float w = (float)(k + col) * 1e-3f;
#else
      const float w = Wsrc[(int64_t)(k < K2 ? k : K2 - 1) * N1 + (col < N1 ? col : N1 - 1)];   // clamped, unconditional
#endif
      // (masked by a multiplication: behind a select the compiler puts every load under its own exec-mask branch with a
      //  vmcnt(0) wait -- 148 dependent round trips, 34 us)
      breg[s_] = w * ((k < K2 && col < N1) ? 1.0f : 0.0f);
    }
  }
  for (int i = tid; i < N1 * N2; i += 64 * AF_NW) w4s[i] = a.W4[i];
  af_build(a, xs0, LD, Kp, row0, row_end, tid, 64 * AF_NW);
  __syncthreads();
  for (int m = 0; m < ntile; ++m) {
    float* xs = (m & 1) ? xs1 : xs0;
    if (wave < ntn) {
      hf_f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
      const int n0[1] = {wave * 16};
      const float* xrow = xs + lc * LD + lq * KQ;
#ifndef AFP_NOMFMA
#ifdef AFP_TWOACC
      hf_f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < KQ; s4 += 4) {
        const float4 av = *reinterpret_cast<const float4*>(xrow + s4);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, breg[s4 + 0], acc[0], 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, breg[s4 + 1], acc2, 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, breg[s4 + 2], acc[0], 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, breg[s4 + 3], acc2, 0, 0, 0);
      }
      acc[0] += acc2;
#else
#pragma unroll
      for (int s4 = 0; s4 < KQ; s4 += 4) {
        const float4 av = *reinterpret_cast<const float4*>(xrow + s4);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, breg[s4 + 0], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, breg[s4 + 1], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, breg[s4 + 2], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, breg[s4 + 3], acc[0], 0, 0, 0);
      }
#endif
#else
      acc[0][0] = xrow[0] * breg[0] + breg[KQ - 1];
#endif
      const int col = n0[0] + lc;
      if (col < N1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = m * 16 + lq * 4 + r;
          const int g = row0 + i;
          if (g < row_end) {
#ifdef AFP_NOEPI
            const float v = fmaxf(acc[0][r], 0.f);
            a1s[i * L1 + col] = v;
#else
            const float v = fmaxf(acc[0][r] + a.qz[(int64_t)(g / T) * N1 + col], 0.f);
            a1s[i * L1 + col] = v;
            a.a1[(int64_t)g * N1 + col] = v;
#endif
          }
        }
      }
    } else if (m + 1 < ntile) {
#ifndef AFP_NOBUILD
      af_build(a, (m & 1) ? xs0 : xs1, LD, Kp, row0 + (m + 1) * 16, row_end, tid - 64 * ntn, 64 * (AF_NW - ntn));
#endif
    }
    __syncthreads();
  }

#ifdef AFP_NOTAIL
  return;
#endif
  // the tail (attn_tail_fwd_kernel's arithmetic, same order): the workgroup's S samples side by side, 512 / S threads each
  const int gsz = 64 * AF_NW / a.S, s_ = tid / gsz, gt = tid - s_ * gsz;
  const bool live = s_ < ns;
  const int b = b0 + (live ? s_ : 0);
  const float* a1b = a1s + s_ * T * L1;
  float* a2l = a2s + s_ * T * L2;
  float* sc = scs + s_ * T;
  if (live) {
    float* a2b = a.a2 + (int64_t)b * T * N2;
    for (int i = gt; i < T * N2; i += gsz) {
      const int t = i / N2, n = i - t * N2;
      float acc = 0.f;
#pragma unroll 8
      for (int k = 0; k < N1; ++k) acc = fmaf(a1b[t * L1 + k], w4s[k * N2 + n], acc);
      const float v = fmaxf(acc + a.b4[n], 0.f);
      a2l[t * L2 + n] = v;
      a2b[i] = v;
    }
  }
  __syncthreads();
  if (live) {
    const int len = a.length[b];
    for (int t = gt; t < T; t += gsz) {
      float acc = 0.f;
      for (int n = 0; n < N2; ++n) acc = fmaf(a2l[t * L2 + n], a.w5[n], acc);
      sc[t] = t < len ? acc + a.b5[0] : -4294967295.0f;
    }
  }
  __syncthreads();
  float mx = -INFINITY, den = 0.f;
  if (live) {
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, sc[t]);
    for (int t = 0; t < T; ++t) den += expf(sc[t] - mx);
  }
  __syncthreads();
  if (live) {
    for (int t = gt; t < T; t += gsz) {
      const float v = expf(sc[t] - mx) / den;
      sc[t] = v;
      a.score[(int64_t)b * T + t] = v;
    }
  }
  __syncthreads();
  if (live) {
    for (int j = gt; j < 2 * H; j += gsz) {
      const bool us = j < H;
      const float* rep = (us ? a.ur : a.ir) + (int64_t)b * T * H + (us ? j : j - H);
      float acc = 0.f;
      for (int t = 0; t < T; ++t) acc = fmaf(rep[(int64_t)t * H], sc[t], acc);
      const int off = us ? a.off_u : a.off_i;
      if (off >= 0) a.head[(int64_t)b * a.ldh + off + (us ? j : j - H)] = acc;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Backward of the folded dense_3 into the attention's input rows, and of the rows into the recurrent states, in ONE
// launch: d inp = da1 . Weff^T ([B*T, 2Dk], K = 80) on v_mfma_f32_16x16x4_f32 with Weff^T resident in registers, then
//   d k[b,t]  = d inp[:, :Dk] + q[b] o d inp[:, Dk:]  (+ the pooled-state path score[b,t] * d head into the two states)
//   d q[b]    = sum_t k[b,t] o d inp[:, Dk:]
// straight from the tile in LDS.  As two launches (an N = 592, K = 80 bf16x3 product of 51 us whose 43 MB result
// attn_inp_bwd_kernel read back, 22 us) the [B*T, 2Dk] matrix went out to HBM and came straight back.
constexpr int AB_NTW = 5;          // 16-column tiles of d inp per wave (8 waves: 2Dk <= 640)
constexpr int AB_KQ = 20;          // k-steps per lane quarter: N1 = 80

struct AttnBwdArgs {
  int B, T, H, NI, S;
  const float* da1; const float* Weff; const float* q; const float* ur; const float* ir; const float* info;
  const float* score; const float* dhead; int ldh, off_u, off_i;
  float* dur; float* dir; float* dinfo; float* dq;
  // pool != 0: the pooling / masked softmax / dense_5 / dense_4 backward of the samples runs here first
  // (attn_pool_bwd_kernel's arithmetic) and da1 is an OUTPUT, like ds and da2
  int pool, N2;
  const float* a2; const float* a1; const float* w5; const float* W4; const int32_t* length;
  float* ds; float* da2; float* da1_out;
};

__global__ __launch_bounds__(64 * AF_NW) void attn_inp_bwd_fused_kernel(const AttnBwdArgs a) {
  extern __shared__ float sm[];
  const int T = a.T, H = a.H, NI = a.NI;
  const int Dk = 2 * H + NI, Dk4 = Dk >> 2, K2 = 2 * Dk, N1 = 4 * AB_KQ;
  const int LDA = N1 + 4, LDO = K2 + 4;
  const int RT = ((a.S * T + 15) >> 4) << 4;      // rows of the workgroup, padded to whole M-tiles
  float* da1s = sm;                       // [RT][LDA]  da1 rows of the workgroup's samples
  float* ot = da1s + RT * LDA;            // [16][LDO]  d inp tile, then (first half) k o d inp[:, Dk:]
  float* dqs = ot + 16 * LDO;             // [S][Dk]
  float* sds = dqs + a.S * Dk;            // [S][T]            (pool)
  float* d2s = sds + a.S * T;             // [S][T][N2 + 1]    (pool)
  float* w4s = d2s + a.S * T * (a.N2 + 1);  // [N1][N2 + 1]   (pool)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, lq = lane >> 4;
  const int b0 = blockIdx.x * a.S;
  const int ns = min(a.S, a.B - b0);
  const int row0 = b0 * T, row_end = row0 + ns * T;
  const int ntile = (ns * T + 15) >> 4;
  const int ntn = (K2 + 15) >> 4;         // <= 8 * AB_NTW (checked by the launcher)

  // B operands: B[k][n] = Weff[n][k], k contiguous -- five 16-byte loads per tile and lane, unconditional (clamped
  // row, masked by a multiplication: see attn_fwd_fused_kernel)
  float breg[AB_NTW][AB_KQ];
#pragma unroll
  for (int t = 0; t < AB_NTW; ++t) {
    const int n = (wave + AF_NW * t) * 16 + lc;
    const float msk = n < K2 ? 1.0f : 0.0f;
    const float* wrow = a.Weff + (int64_t)(n < K2 ? n : K2 - 1) * N1 + lq * AB_KQ;
#pragma unroll
    for (int s4 = 0; s4 < AB_KQ; s4 += 4) {
      const float4 w = ld4(wrow + s4);
      breg[t][s4 + 0] = w.x * msk; breg[t][s4 + 1] = w.y * msk; breg[t][s4 + 2] = w.z * msk; breg[t][s4 + 3] = w.w * msk;
    }
  }
  for (int i = tid; i < a.S * Dk; i += 64 * AF_NW) dqs[i] = 0.f;
  for (int i = tid; i < RT * LDA; i += 64 * AF_NW) da1s[i] = 0.f;
  __syncthreads();
  if (!a.pool) {
    // da1 rows of the workgroup's samples (computed by an earlier launch)
    for (int e = tid; e < ns * T * (N1 / 4); e += 64 * AF_NW) {
      const int i = e / (N1 / 4), c = (e - i * (N1 / 4)) * 4;
      *reinterpret_cast<float4*>(da1s + i * LDA + c) = ld4(a.da1 + (int64_t)(row0 + i) * N1 + c);
    }
  } else {
    // attn_pool_bwd_kernel's part, the S samples side by side (512 / S threads each):
    //   dscore_t = duf.ur_t + dif.ir_t ; ds_t = score_t (dscore_t - sum score*dscore) [t < len]
    //   da2[t][n] = ds_t * w5[n] * [a2 > 0] ; da1[t][k] = [a1 > 0] sum_n da2[t][n] W4[k][n]
    const int NA = a.N2, LW = NA + 1;
    const int gsz = 64 * AF_NW / a.S, s_ = tid / gsz, gt = tid - s_ * gsz;
    const bool live = s_ < ns;
    const int b = b0 + (live ? s_ : 0);
    float* sd = sds + s_ * T;
    float* d2 = d2s + s_ * T * LW;
    for (int i = tid; i < N1 * NA; i += 64 * AF_NW) { const int k = i / NA; w4s[k * LW + (i - k * NA)] = a.W4[i]; }
    constexpr int LPT = 4;                           // lanes per slice of the dscore dot products
    if (live) {
      const int gl = gt % LPT;
      const float* du = a.off_u >= 0 ? a.dhead + (int64_t)b * a.ldh + a.off_u : nullptr;
      const float* di = a.off_i >= 0 ? a.dhead + (int64_t)b * a.ldh + a.off_i : nullptr;
      for (int t0 = 0; t0 < T; t0 += gsz / LPT) {
        const int t = t0 + gt / LPT;
        const int64_t bt = (int64_t)b * T + (t < T ? t : T - 1);
        float part = 0.f;
        for (int j = gl; j < H; j += LPT) {
          if (du) part = fmaf(du[j], a.ur[bt * H + j], part);
          if (di) part = fmaf(di[j], a.ir[bt * H + j], part);
        }
        part = group_sum(part, LPT);
        if (gl == 0 && t < T) sd[t] = part;
      }
    }
    __syncthreads();
    if (live) {
      const int len = a.length[b];
      float tot = 0.f;
      for (int t = 0; t < T; ++t) tot = fmaf(a.score[(int64_t)b * T + t], sd[t], tot);
      for (int i = gt; i < T * NA; i += gsz) {
        const int t = i / NA, n = i - t * NA;
        const int64_t bt = (int64_t)b * T + t;
        const float g = t < len ? a.score[bt] * (sd[t] - tot) : 0.f;
        const float dv = a.a2[bt * NA + n] > 0.f ? g * a.w5[n] : 0.f;
        a.da2[bt * NA + n] = dv;
        d2[t * LW + n] = dv;
        if (n == 0) a.ds[bt] = g;
      }
    }
    __syncthreads();
    if (live) {
      for (int i = gt; i < T * N1; i += gsz) {
        const int t = i / N1, k = i - t * N1;
        const int64_t e = ((int64_t)b * T + t) * N1 + k;
        float acc = 0.f;
#pragma unroll 8
        for (int n = 0; n < NA; ++n) acc = fmaf(d2[t * LW + n], w4s[k * LW + n], acc);
        const float v = a.a1[e] > 0.f ? acc : 0.f;
        a.da1_out[e] = v;
        da1s[(s_ * T + t) * LDA + k] = v;
      }
    }
  }
  __syncthreads();

  for (int m = 0; m < ntile; ++m) {
    const int g0 = row0 + m * 16;
    {
      float4 av[AB_KQ / 4];
#pragma unroll
      for (int s4 = 0; s4 < AB_KQ / 4; ++s4)
        av[s4] = *reinterpret_cast<const float4*>(da1s + (m * 16 + lc) * LDA + lq * AB_KQ + 4 * s4);
#pragma unroll
      for (int t = 0; t < AB_NTW; ++t) {
        const int n0 = (wave + AF_NW * t) * 16;
        if (n0 >= ntn * 16) continue;                   // (wave-uniform)
        hf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < AB_KQ / 4; ++s4) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4].x, breg[t][4 * s4 + 0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4].y, breg[t][4 * s4 + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4].z, breg[t][4 * s4 + 2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4].w, breg[t][4 * s4 + 3], acc, 0, 0, 0);
        }
        const int col = n0 + lc;
        if (col < K2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) ot[(lq * 4 + r) * LDO + col] = acc[r];
        }
      }
    }
    __syncthreads();
    // d k rows out; k o d inp[:, Dk:] left in the tile's first half for the d q sums
    constexpr int U = 3;
    for (int e0 = tid; e0 < 16 * Dk4; e0 += U * 64 * AF_NW) {
      float4 kv[U], qq[U], pl[U];
      float sc[U];
      int ii[U], jj[U], gg[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * 64 * AF_NW;
        const int ec = e < 16 * Dk4 ? e : 0;
        ii[u] = ec / Dk4; jj[u] = (ec - ii[u] * Dk4) * 4;
        gg[u] = g0 + ii[u];
        ok[u] = e < 16 * Dk4 && gg[u] < row_end;
        const int g = gg[u] < row_end ? gg[u] : row_end - 1;
        const int j = jj[u], b = g / T;
        const float* src = j < H ? a.ur + (int64_t)g * H + j : (j < 2 * H ? a.ir + (int64_t)g * H + (j - H) : a.info + (int64_t)g * NI + (j - 2 * H));
        kv[u] = ld4(src);
        qq[u] = ld4(a.q + (int64_t)b * Dk + j);
        sc[u] = a.score[g];
        // pooled-state gradient of the column's state (none for the atten_info columns / a side the head does not take)
        const int off = j < H ? a.off_u : a.off_i;
        const bool hasp = j < 2 * H && off >= 0;
        const float4 p4 = ld4(a.dhead + (int64_t)b * a.ldh + (hasp ? off + (j < H ? j : j - H) : 0));
        const float pm = hasp ? 1.0f : 0.0f;
        pl[u] = make_float4(p4.x * pm, p4.y * pm, p4.z * pm, p4.w * pm);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (e0 + u * 64 * AF_NW >= 16 * Dk4) continue;
        float* o1 = ot + ii[u] * LDO + jj[u];
        const float4 d1 = *reinterpret_cast<const float4*>(o1);
        const float4 d3 = *reinterpret_cast<const float4*>(o1 + Dk);
        float4 pr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok[u]) {
          const int j = jj[u];
          float4 o;
          o.x = fmaf(d3.x, qq[u].x, d1.x) + pl[u].x * sc[u]; o.y = fmaf(d3.y, qq[u].y, d1.y) + pl[u].y * sc[u];
          o.z = fmaf(d3.z, qq[u].z, d1.z) + pl[u].z * sc[u]; o.w = fmaf(d3.w, qq[u].w, d1.w) + pl[u].w * sc[u];
          float* dst = j < H ? a.dur + (int64_t)gg[u] * H + j
                             : (j < 2 * H ? a.dir + (int64_t)gg[u] * H + (j - H) : a.dinfo + (int64_t)gg[u] * NI + (j - 2 * H));
          st4(dst, o);
          pr = make_float4(d3.x * kv[u].x, d3.y * kv[u].y, d3.z * kv[u].z, d3.w * kv[u].w);
        }
        *reinterpret_cast<float4*>(o1) = pr;
      }
    }
    __syncthreads();
    // d q[s][j] += sum over the tile's rows of sample s, in row order
    for (int e = tid; e < ns * Dk4; e += 64 * AF_NW) {
      const int s_ = e / Dk4, j = (e - s_ * Dk4) * 4;
      const int lo = max(0, s_ * T - m * 16), hi = min(16, (s_ + 1) * T - m * 16);
      if (lo < hi) {
        float4 acc = *reinterpret_cast<const float4*>(dqs + s_ * Dk + j);
        for (int i = lo; i < hi; ++i) {
          const float4 v = *reinterpret_cast<const float4*>(ot + i * LDO + j);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(dqs + s_ * Dk + j) = acc;
      }
    }
    __syncthreads();
  }
  for (int e = tid; e < ns * Dk4; e += 64 * AF_NW) {
    const int s_ = e / Dk4, j = (e - s_ * Dk4) * 4;
    st4(a.dq + (int64_t)(b0 + s_) * Dk + j, *reinterpret_cast<const float4*>(dqs + s_ * Dk + j));
  }
}


// ---------------------------------------------------------------------------------------------------------------
// build_fc_net backward down to the head's input in ONE launch (score.py:68-76 backwards): from dz2 (written by the
// fused forward) a workgroup of 16 samples computes dz1 = [f1 > 0] (dz2 . W2^T) / keep and d bn1 = dz1 . W1^T on
// v_mfma_f32_16x16x4_f32, then bn1's backward (d head = d bn * gamma * rs, the d gamma terms d bn * x * rs).  The
// transposed weights are k-contiguous per output column, so a lane takes its B operands with 16-byte loads: fc2's two
// tiles per wave stay in registers, fc1's are streamed a tile ahead.  It was three launches (two K = 80 / K = 200
// products of 1024 rows and the element-wise bn1 backward: 14 + 13 + 6 us).
constexpr int HB_K2Q = 20;      // fc2: K = FC2 = 80 -> k-steps per lane quarter
constexpr int HB_K1Q = 52;      // fc1: K = FC1 = 200 -> padded to 208

struct HeadBwdArgs {
  int B, Dh;
  const float* dz2; const float* W2; const float* f1; float keep;
  const float* W1; const float* x; const float* gamma; float rs;
  float* dz1; float* dbn; float* dhead; float* tmp;
};

__global__ __launch_bounds__(64 * HF_NW) void head_bwd_fused_kernel(const HeadBwdArgs a) {
  constexpr int N1 = 4 * HB_K2Q * 0 + 200, N2 = 80, KP1 = 4 * HB_K1Q;     // FC1, FC2, FC1 padded
  constexpr int LD2 = N2 + 4, LD1 = KP1 + 4;
  __shared__ __attribute__((aligned(16))) float z2s[HF_ROWS * LD2];          // dz2 rows
  __shared__ __attribute__((aligned(16))) float z1s[HF_ROWS * LD1];          // dz1 rows, zero beyond FC1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, lq = lane >> 4;
  const int b0 = blockIdx.x * HF_ROWS;
  for (int e = tid; e < HF_ROWS * (N2 / 4); e += 64 * HF_NW) {
    const int i = e / (N2 / 4), c = (e - i * (N2 / 4)) * 4;
    const int row = b0 + i < a.B ? b0 + i : a.B - 1;
    float4 v = ld4(a.dz2 + (int64_t)row * N2 + c);
    if (b0 + i >= a.B) v = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(z2s + i * LD2 + c) = v;
  }
  for (int e = tid; e < HF_ROWS * (LD1 - N1); e += 64 * HF_NW) {
    const int i = e / (LD1 - N1), c = e - i * (LD1 - N1);
    z1s[i * LD1 + N1 + c] = 0.f;
  }
  // fc2 backward: tiles wave and wave + 8 of the 13; B[k][n] = W2[n][k]
  float b2[2][HB_K2Q];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n = (wave + HF_NW * t) * 16 + lc;
    const float msk = n < N1 ? 1.0f : 0.0f;
    const float* wrow = a.W2 + (int64_t)(n < N1 ? n : N1 - 1) * N2 + lq * HB_K2Q;
#pragma unroll
    for (int s4 = 0; s4 < HB_K2Q; s4 += 4) {
      const float4 w = ld4(wrow + s4);
      b2[t][s4 + 0] = w.x * msk; b2[t][s4 + 1] = w.y * msk; b2[t][s4 + 2] = w.z * msk; b2[t][s4 + 3] = w.w * msk;
    }
  }
  __syncthreads();
  {
    float4 av[HB_K2Q / 4];
#pragma unroll
    for (int s4 = 0; s4 < HB_K2Q / 4; ++s4) av[s4] = *reinterpret_cast<const float4*>(z2s + lc * LD2 + lq * HB_K2Q + 4 * s4);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int n0 = (wave + HF_NW * t) * 16;
      if (n0 >= N1) continue;
      hf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < HB_K2Q / 4; ++s4) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4].x, b2[t][4 * s4 + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4].y, b2[t][4 * s4 + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4].z, b2[t][4 * s4 + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4].w, b2[t][4 * s4 + 3], acc, 0, 0, 0);
      }
      const int col = n0 + lc;
      if (col < N1) {
        // (the four mask values first, from clamped rows: a load behind `row < B` sits in its own exec-mask branch with a
        //  vmcnt(0) wait -- four dependent round trips per tile)
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = a.f1[(int64_t)min(b0 + lq * 4 + r, a.B - 1) * N1 + col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = lq * 4 + r, row = b0 + i;
          const float q = acc[r] / a.keep;
          const float v = (row < a.B && y[r] > 0.f) ? q : 0.f;      // relu (+ dropout) of fc1, as the GEMM epilogue had it
          if (row < a.B && blockIdx.y == 0) a.dz1[(int64_t)row * N1 + col] = v;
          z1s[i * LD1 + col] = v;
        }
      }
    }
  }
  __syncthreads();
  // fc1 backward + bn1 backward: tiles wave, wave + 8, ... of ceil(Dh / 16); B[k][n] = W1[n][k], streamed one tile ahead
  // (gridDim.y workgroups share a row tile's column tiles -- each has computed dz1 for itself: at 1024 samples the 64 row
  //  tiles alone left three quarters of the chip idle for 27 us of the launch stream; four shares: the same sums, 15 us)
  const int nt_all = (a.Dh + 15) >> 4;
  const int per = (nt_all + (int)gridDim.y - 1) / (int)gridDim.y;
  const int t0 = (int)blockIdx.y * per, nt = min(nt_all, t0 + per);
  float4 av[HB_K1Q / 4];
#pragma unroll
  for (int s4 = 0; s4 < HB_K1Q / 4; ++s4) av[s4] = *reinterpret_cast<const float4*>(z1s + lc * LD1 + lq * HB_K1Q + 4 * s4);
  float4 bw[2][HB_K1Q / 4];
  auto fetch = [&](float4 (&dst)[HB_K1Q / 4], int tile) {
    const int n = tile * 16 + lc;
    const float* wrow = a.W1 + (int64_t)(n < a.Dh ? n : a.Dh - 1) * N1 + lq * HB_K1Q;
#pragma unroll
    for (int s4 = 0; s4 < HB_K1Q / 4; ++s4) {
      // (the last quarter's k range runs past FC1 = 200 into the next row: those products meet the zero padding of z1s)
      const int k = lq * HB_K1Q + 4 * s4;
      dst[s4] = ld4(k + 3 < N1 ? wrow + 4 * s4 : a.W1);
    }
  };
  if (t0 + wave < nt) fetch(bw[0], t0 + wave);
  int cur = 0;
  for (int tile = t0 + wave; tile < nt; tile += HF_NW, cur ^= 1) {
    if (tile + HF_NW < nt) {
      if (cur == 0) fetch(bw[1], tile + HF_NW); else fetch(bw[0], tile + HF_NW);
    }
    hf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int col = tile * 16 + lc;
    const float cm = col < a.Dh ? 1.0f : 0.0f;
#pragma unroll
    for (int s4 = 0; s4 < HB_K1Q / 4; ++s4) {
      const float4 w = cur == 0 ? bw[0][s4] : bw[1][s4];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4].x, w.x * cm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4].y, w.y * cm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4].z, w.z * cm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4].w, w.w * cm, acc, 0, 0, 0);
    }
    {
      const int cc = col < a.Dh ? col : a.Dh - 1;
      const float gs = a.gamma[cc] * a.rs;
      float xv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) xv[r] = a.x[(int64_t)min(b0 + lq * 4 + r, a.B - 1) * a.Dh + cc];     // clamped, unconditional
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = b0 + lq * 4 + r;
        if (row < a.B && col < a.Dh) {
          const int64_t e = (int64_t)row * a.Dh + col;
          const float dy = acc[r];
          a.dbn[e] = dy;
          a.dhead[e] = dy * gs;
          a.tmp[e] = dy * (xv[r] * a.rs);
        }
      }
    }
  }
}

}  // namespace

// (what score_backward asks to know whether the forward pass has left dz2 behind)
bool score_head_fwd_fused_fits(int B, int Dh, int N1, int N2) {
  const int LD0 = ((Dh + 15) & ~15) + 4, LD1 = ((N1 + 15) & ~15) + 4, LD2 = ((N2 + 15) & ~15) + 4;
  return B > 0 && (size_t)HF_ROWS * (LD0 + LD1 + LD2) * sizeof(float) <= 150 * 1024;
}

// Returns SCORE_E_SHAPE when the shape does not fit the fused kernel (the caller then runs the layer-by-layer path).
int score_launch_head_fwd_fused(int B, int Dh, int N1, int N2, const float* x, const float* gamma, const float* beta,
                                float rs, const float* W1, const float* b1, const float* W2, const float* b2,
                                const float* W3, const float* b3, float keep, const uint8_t* mask0, const uint8_t* mask1,
                                uint64_t seed0, uint64_t seed1, const int32_t* label, float* bn, float* f1, float* f2,
                                float* logit, float* y, float* lossb, float* dlogit, int Bglobal, hipStream_t s,
                                const uint64_t* seed_dev, float* dz2, int single_launch) {
  const int LD0 = ((Dh + 15) & ~15) + 4, LD1 = ((N1 + 15) & ~15) + 4, LD2 = ((N2 + 15) & ~15) + 4;
  const size_t lds = (size_t)HF_ROWS * (LD0 + LD1 + LD2) * sizeof(float);
  if (!score_head_fwd_fused_fits(B, Dh, N1, N2)) return SCORE_E_SHAPE;
  static thread_local bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(head_fwd_fused_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  HeadFwdArgs a;
  a.B = B; a.Dh = Dh; a.N1 = N1; a.N2 = N2; a.Bglobal = Bglobal;
  a.x = x; a.gamma = gamma; a.beta = beta; a.rs = rs;
  a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.W3 = W3; a.b3 = b3;
  a.keep = keep; a.drop = keep < 1.f ? 1 : 0; a.mask0 = mask0; a.mask1 = mask1; a.seed0 = seed0; a.seed1 = seed1; a.seed_dev = seed_dev;
  a.label = label; a.bn = bn; a.f1 = f1; a.f2 = f2; a.logit = logit; a.y = y; a.lossb = lossb; a.dlogit = dlogit;
  a.dz2 = dz2;
  const int mt = (B + HF_ROWS - 1) / HF_ROWS, nt1 = (N1 + 15) >> 4;
  // two launches from 32 row tiles up (one launch: score_state_t.debug_flags bit 1)
  const bool split = !single_launch && nt1 >= 4 && mt >= 32;
  if (!split) {
    a.phase = 0;
    hipLaunchKernelGGL(head_fwd_fused_kernel, dim3(mt), dim3(64 * HF_NW), lds, s, a);
    SCORE_CHECK_LAUNCH();
    return 0;
  }
  a.phase = 1;
  hipLaunchKernelGGL(head_fwd_fused_kernel, dim3(mt, 4), dim3(64 * HF_NW), lds, s, a);
  SCORE_CHECK_LAUNCH();
  a.phase = 2;
  hipLaunchKernelGGL(head_fwd_fused_kernel, dim3(mt), dim3(64 * HF_NW), lds, s, a);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// Returns SCORE_E_SHAPE when the shape does not fit (the caller then runs build + GEMM + tail as separate launches).
int score_launch_attn_fwd_fused(int B, int T, int H, int NI, int N1, int N2, const float* q, const float* ur,
                                const float* ir, const float* info, const float* Weff, const float* qz, const float* W4,
                                const float* b4, const float* w5, const float* b5, const int32_t* length, float* inp,
                                float* a1, float* a2, float* score, float* head, int ldh, int off_u, int off_i,
                                hipStream_t s, int weff_copies, int64_t weff_copy_stride) {
  if (B <= 0 || T <= 0 || (H & 3) || (NI & 3) || N1 > 16 * 5 || N1 <= 0 || N2 <= 0 || (int64_t)B * T >= (1LL << 30))
    return SCORE_E_SHAPE;
  const int K2 = 2 * (2 * H + NI), Kp = (K2 + 15) & ~15, LD = Kp + 4;
  // samples per workgroup: enough workgroups for the chip first, then fuller M-tiles (4 samples of 18 slices: 72 rows in
  // five tiles); bounded by LDS
  int S = B >= 4 * 256 ? 4 : B >= 2 * 256 ? 2 : 1;
  size_t lds = 0;
  for (; S >= 1; S >>= 1) {
    lds = ((size_t)2 * 16 * LD + (size_t)S * T * (N1 + 1) + (size_t)N1 * N2 + (size_t)S * T * (N2 + 1) + (size_t)S * T) * sizeof(float);
    if (lds <= 150 * 1024) break;
  }
  if (S < 1) return SCORE_E_SHAPE;
  const int kq = Kp / 4;
  if (kq != 148 && kq != 52 && kq != 44 && kq != 36 && kq != 84) return SCORE_E_SHAPE;    // the instantiated register-resident widths
  static thread_local bool attr_set = false;
  if (!attr_set) {
    for (const void* f : {reinterpret_cast<const void*>(attn_fwd_fused_kernel<148>),
                          reinterpret_cast<const void*>(attn_fwd_fused_kernel<52>),
                          reinterpret_cast<const void*>(attn_fwd_fused_kernel<44>),
                          reinterpret_cast<const void*>(attn_fwd_fused_kernel<36>),
                          reinterpret_cast<const void*>(attn_fwd_fused_kernel<84>)}) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      if (e != hipSuccess) return (int)e;
    }
    attr_set = true;
  }
  AttnFwdArgs a;
  a.B = B; a.T = T; a.H = H; a.NI = NI; a.N1 = N1; a.N2 = N2; a.S = S;
  a.q = q; a.ur = ur; a.ir = ir; a.info = info; a.Weff = Weff; a.qz = qz;
  a.W4 = W4; a.b4 = b4; a.w5 = w5; a.b5 = b5; a.length = length;
  a.inp = inp; a.a1 = a1; a.a2 = a2; a.score = score; a.head = head; a.ldh = ldh; a.off_u = off_u; a.off_i = off_i;
  a.wcopies = weff_copies > 0 ? weff_copies : 1; a.wstride = weff_copy_stride;
  const dim3 grid((B + S - 1) / S), block(64 * AF_NW);
  switch (Kp / 4) {
    case 148: hipLaunchKernelGGL(attn_fwd_fused_kernel<148>, grid, block, lds, s, a); break;   // H = 128, K = 10
    case 52: hipLaunchKernelGGL(attn_fwd_fused_kernel<52>, grid, block, lds, s, a); break;     // H = 32, K = 10
    case 44: hipLaunchKernelGGL(attn_fwd_fused_kernel<44>, grid, block, lds, s, a); break;     // H = 32, K = 5
    case 36: hipLaunchKernelGGL(attn_fwd_fused_kernel<36>, grid, block, lds, s, a); break;     // H = 16, K = 10
    case 84: hipLaunchKernelGGL(attn_fwd_fused_kernel<84>, grid, block, lds, s, a); break;     // H = 64, K = 10
    default: return SCORE_E_SHAPE;
  }
  SCORE_CHECK_LAUNCH();
  return 0;
}

static bool ab_plan(int B, int T, int H, int NI, int N1, int N2, int ldh, int off_u, int off_i, bool pool, int* S_out,
                    size_t* lds_out) {
  const int Dk = 2 * H + NI, K2 = 2 * Dk;
  if (B <= 0 || T <= 0 || (H & 3) || (NI & 3) || N1 != 4 * AB_KQ || K2 > 16 * AF_NW * AB_NTW || (int64_t)B * T >= (1LL << 30) ||
      (ldh & 3) || (off_u >= 0 && (off_u & 3)) || (off_i >= 0 && (off_i & 3)) || (pool && N2 <= 0))
    return false;
  const int S = B >= 4 * 256 ? 4 : B >= 2 * 256 ? 2 : 1;
  const int RT = ((S * T + 15) >> 4) << 4;
  const size_t lds = ((size_t)RT * (N1 + 4) + (size_t)16 * (K2 + 4) + (size_t)S * Dk +
                      (pool ? (size_t)S * T + (size_t)S * T * (N2 + 1) + (size_t)N1 * (N2 + 1) : 0)) * sizeof(float);
  if (lds > 150 * 1024) return false;
  *S_out = S; *lds_out = lds;
  return true;
}
bool score_attn_inp_bwd_fused_fits(int B, int T, int H, int NI, int N1, int N2, int ldh, int off_u, int off_i, bool pool) {
  int S; size_t lds;
  return ab_plan(B, T, H, NI, N1, N2, ldh, off_u, off_i, pool, &S, &lds);
}

// Returns SCORE_E_SHAPE when the shape does not fit (the caller then runs the product and attn_inp_bwd_kernel).
int score_launch_attn_inp_bwd_fused(int B, int T, int H, int NI, int N1, const float* da1, const float* Weff, const float* q,
                                    const float* ur, const float* ir, const float* info, const float* score,
                                    const float* dhead, int ldh, int off_u, int off_i, float* dur, float* dir, float* dinfo,
                                    float* dq, hipStream_t s, int N2, const float* a2, const float* a1, const float* w5,
                                    const float* W4, const int32_t* length, float* ds, float* da2, float* da1_out) {
  const int pool = a2 != nullptr;
  if (pool && (!a1 || !w5 || !W4 || !length || !ds || !da2 || !da1_out)) return SCORE_E_BADARG;
  int S; size_t lds;
  if (!ab_plan(B, T, H, NI, N1, N2, ldh, off_u, off_i, pool != 0, &S, &lds)) return SCORE_E_SHAPE;
  static thread_local bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_inp_bwd_fused_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  AttnBwdArgs a;
  a.B = B; a.T = T; a.H = H; a.NI = NI; a.S = S;
  a.da1 = da1; a.Weff = Weff; a.q = q; a.ur = ur; a.ir = ir; a.info = info; a.score = score; a.dhead = dhead;
  a.ldh = ldh; a.off_u = off_u; a.off_i = off_i; a.dur = dur; a.dir = dir; a.dinfo = dinfo; a.dq = dq;
  a.pool = pool; a.N2 = pool ? N2 : 0; a.a2 = a2; a.a1 = a1; a.w5 = w5; a.W4 = W4; a.length = length; a.ds = ds; a.da2 = da2;
  a.da1_out = da1_out;
  hipLaunchKernelGGL(attn_inp_bwd_fused_kernel, dim3((B + S - 1) / S), dim3(64 * AF_NW), lds, s, a);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// Returns SCORE_E_SHAPE when the layer widths are not build_fc_net's (the caller then runs the layer-by-layer path).
int score_launch_head_bwd_fused(int B, int Dh, int N1, int N2, const float* dz2, const float* W2, const float* f1, float keep,
                                const float* W1, const float* x, const float* gamma, float rs, float* dz1, float* dbn,
                                float* dhead, float* tmp, hipStream_t s) {
  if (B <= 0 || Dh <= 0 || N1 != 200 || N2 != 80) return SCORE_E_SHAPE;
  HeadBwdArgs a;
  a.B = B; a.Dh = Dh; a.dz2 = dz2; a.W2 = W2; a.f1 = f1; a.keep = keep; a.W1 = W1; a.x = x; a.gamma = gamma; a.rs = rs;
  a.dz1 = dz1; a.dbn = dbn; a.dhead = dhead; a.tmp = tmp;
  // few row tiles and many column tiles of d bn1: the column tiles dealt to up to four workgroups per row tile
  const int mt = (B + HF_ROWS - 1) / HF_ROWS, ntile = (Dh + 15) >> 4;
  const int ny = (mt <= 128 && ntile >= 4 * HF_NW) ? (mt <= 64 ? 4 : 2) : 1;
  hipLaunchKernelGGL(head_bwd_fused_kernel, dim3(mt, ny), dim3(64 * HF_NW), 0, s, a);
  SCORE_CHECK_LAUNCH();
  return 0;
}
