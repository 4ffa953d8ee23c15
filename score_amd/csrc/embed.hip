// Embedding-table kernels: row gather, fused gather + co-attention forward/backward,
// target-row gather/backward.  HBM-bound: every lane moves 16 B per load, all K
// neighbour rows of a unit are issued before the first use.
//
// Mapping ("slot" = one float4 of the F*D-wide concatenated feature vector of a
// neighbour, score.py:51-66 reshape): a GROUP of GS lanes (power of two, <= 64)
// owns one (b,t) unit; lane gl owns slots gl, gl+GS, ... (SPL of them) and walks
// the K neighbours in registers, so sums over K are sequential per lane (bitwise
// reproducible) and only the K relateness scores cross lanes.
#include <string.h>
#include "common.h"
#include "kernels.h"

// ------------------------------------------------------------------ plain gather (score.py:51-66)
__global__ void gather_rows_kernel(const float* __restrict__ table, int D4, const int32_t* __restrict__ idx,
                                   int64_t n_chunks, float* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n_chunks; i += stride) {
    int64_t r = i / D4;
    int c = (int)(i - r * D4);
    int64_t row = idx[r];
    st4(out + i * 4, ld4(table + (row * D4 + c) * 4));
  }
}

extern "C" int score_gather_fwd(const float* table, int64_t n_rows, int32_t D, const int32_t* idx,
                                int64_t n_idx, float* out, void* stream) {
  if (!table || !idx || !out || n_rows <= 0 || n_idx < 0) return SCORE_E_BADARG;
  if (D <= 0 || (D & 3)) return SCORE_E_SHAPE;
  if (n_idx == 0) return 0;
  int64_t n_chunks = n_idx * (D / 4);
  int blocks = (int)(cdiv64(n_chunks, 256) < 4096 ? cdiv64(n_chunks, 256) : 4096);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, D / 4, idx,
                     n_chunks, out);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ fused gather + co-attention
// One launch serves both co_attention calls of the model (score.py:196-197): blocks
// [0, first_block[1]) work on call 0, the rest on call 1, each with its own geometry.
static inline void coattn_geom(int D, int F, int* GS, int* SPL, int* nslots) {
  *nslots = F * (D / 4);
  int gs = 1;
  while (gs < *nslots && gs < 64) gs <<= 1;
  *GS = gs;
  *SPL = (*nslots + gs - 1) / gs;
}

template <int KMAX, int SPL>
__global__ __launch_bounds__(256) void coattn_fwd_kernel(const CoattnArgs a) {
  const int ci = (int)blockIdx.x >= a.c[1].first_block ? 1 : 0;
  const CoattnCall& cc = a.c[ci];
  const int GS = cc.GS, nslots = cc.nslots, F = cc.F, K = a.K, D4 = a.D4;
  const int lane = threadIdx.x & 63;
  const int wave = (((int)blockIdx.x - cc.first_block) * (int)blockDim.x + (int)threadIdx.x) >> 6;
  const int upw = 64 / GS;
  const int gl = lane & (GS - 1);
  const int unit = wave * upw + lane / GS;
  const bool unit_ok = unit < a.n_units;
  const int u = unit_ok ? unit : 0;  // clamp: inactive groups still take part in shuffles
  const int Dx = nslots * 4;
  const int D = D4 * 4;
  const float* __restrict__ table = a.table;
  const float* __restrict__ W = cc.W;
  const int mode = a.mode;

  bool ok[SPL];
  int f[SPL], coff[SPL];
  float4 w1[SPL], w2[SPL];
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    int s = gl + j * GS;
    ok[j] = unit_ok && s < nslots;
    int sc = s < nslots ? s : 0;
    f[j] = sc / D4;
    coff[j] = (sc - f[j] * D4) * 4;
    if (mode == 0) {
      w1[j] = ld4(W + Dx + sc * 4);
      w2[j] = ld4(W + 2 * Dx + sc * 4);
    } else {
      w1[j] = w2[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }

  // unit u = b * T + t over the ACTIVE time slices; the index tensors keep their [B, Tidx, K, F] strides
  const int b_idx = u / a.T;
  const int64_t ui = (int64_t)b_idx * a.Tidx + (u - b_idx * a.T);
  const int32_t* __restrict__ i1 = cc.idx1 + ui * K * F;
  const int32_t* __restrict__ i2 = cc.idx2 + ui * K * F;
  // Every load of the unit is unconditional and goes out before anything is consumed: first the 2K row ids,
  // then the 2K rows (16 B per lane each).  Inactive lanes / slots and k >= K read a valid clamped address and
  // are zeroed by a select afterwards: with the loads under `if (k < K)` / `if (ok)` branches the compiler
  // emitted one region per k that waited (vmcnt(0)) for its own two rows before the next k's ids were even
  // requested -- 2K dependent round trips per wave instead of two (0.134 -> 0.097 ms at cfg-3).
  float4 v1[SPL][KMAX];
  float4 sum2[SPL];
  float part[KMAX];
  int32_t ra[SPL][KMAX], rb[SPL][KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int kc = k < K ? k : K - 1;
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
      ra[j][k] = i1[kc * F + f[j]];
      rb[j][k] = i2[kc * F + f[j]];
    }
  }
  // ids outside the table: read as the dummy row, reported once per lane that saw one (clamped positions repeat real
  // ids of the tensor, so nothing is reported that the feed does not hold)
  const uint32_t NR = a.n_rows;
  bool bad1 = false, bad2 = false;
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
      const bool b1 = (uint32_t)ra[j][k] >= NR, b2 = (uint32_t)rb[j][k] >= NR;
      bad1 |= b1; bad2 |= b2;
      ra[j][k] = b1 ? 0 : ra[j][k];
      rb[j][k] = b2 ? 0 : rb[j][k];
    }
  if (a.id_status && (bad1 || bad2)) atomicOr(a.id_status, (bad1 ? 1 << cc.bit1 : 0) | (bad2 ? 1 << cc.bit2 : 0));
  float4 yv[SPL][KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
      v1[j][k] = ld4(table + (int64_t)ra[j][k] * D + coff[j]);
      yv[j][k] = ld4(table + (int64_t)rb[j][k] * D + coff[j]);
    }
#pragma unroll
  for (int j = 0; j < SPL; ++j) sum2[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    part[k] = 0.f;
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
      const bool live = ok[j] && k < K;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 x = live ? v1[j][k] : z;
      const float4 y = live ? yv[j][k] : z;
      v1[j][k] = x;
      sum2[j] = add4(sum2[j], y);
      part[k] += dot4(x, w1[j]) + dot4(y, w2[j]);
    }
  }

  if (mode == 1) {  // RCA: reduce_sum over K (score.py:266-269)
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
      if (!ok[j]) continue;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) o = add4(o, v1[j][k]);
      st4(cc.out1 + (int64_t)u * cc.ld1 + (gl + j * GS) * 4, o);
      st4(cc.out2 + (int64_t)u * cc.ld2 + (gl + j * GS) * 4, sum2[j]);
    }
    return;
  }

  // c = w_t . target + bias (constant over t and i), then r_i = relu(part_i + c)
  float cpart = 0.f;
  if (cc.tidx) {
    // (round 6: straight from the table by the target's ids -- the copy that target_fwd_kernel gathers for the head and the query
    //  branch is the same bits, and reading it put that launch and a stream boundary in front of this kernel, the first of the
    //  step's chain; an id outside the table is the dummy row here as there, and reported there)
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
      uint32_t tr = (uint32_t)cc.tidx[(int64_t)b_idx * F + f[j]];
      tr = tr >= NR ? 0u : tr;
      const float4 tv = ld4(table + (int64_t)tr * D + coff[j]);
      if (ok[j]) cpart += dot4(tv, ld4(W + (gl + j * GS) * 4));
    }
  } else {
#pragma unroll
    for (int j = 0; j < SPL; ++j)
      if (ok[j]) cpart += dot4(ld4(cc.tgt + (int64_t)b_idx * cc.ldt + (gl + j * GS) * 4), ld4(W + (gl + j * GS) * 4));
  }
  float red[KMAX + 1];          // the K partial scores and the target term: reduced across the group together
#pragma unroll
  for (int k = 0; k < KMAX; ++k) red[k] = (k < K) ? part[k] : 0.f;
  red[KMAX] = cpart;
  group_sum_n<KMAX + 1>(red, GS);
  const float c = red[KMAX] + cc.bias[0];
  float r[KMAX];
  float rmax = 0.f, rsum = 0.f;  // relu output >= 0
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    r[k] = 0.f;
    if (k < K) {
      r[k] = fmaxf(red[k] + c, 0.f);
      rmax = fmaxf(rmax, r[k]);
      rsum += r[k];
    }
  }
  float p[KMAX];
  float den = 0.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    p[k] = (k < K) ? expf(r[k] - rmax) : 0.f;
    den += p[k];
  }
  const float inv_den = 1.0f / den;
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    if (!ok[j]) continue;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) o = fma4(p[k] * inv_den, v1[j][k], o);
    st4(cc.out1 + (int64_t)u * cc.ld1 + (gl + j * GS) * 4, o);
    const float fk = (float)K;
    st4(cc.out2 + (int64_t)u * cc.ld2 + (gl + j * GS) * 4,
        make_float4(sum2[j].x / fk, sum2[j].y / fk, sum2[j].z / fk, sum2[j].w / fk));
  }
  // atten_info = [K*r_0..K*r_{K-1}, sum_i r_i (K times)]  (score.py:165-166)
  if (unit_ok) {
    for (int i = gl; i < 2 * K; i += GS) {
      float val = rsum;
      float rv = 0.f;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (i == k) rv = r[k];
      if (i < K) {
        val = (float)K * rv;
        cc.rsave[(int64_t)u * K + i] = rv;
      }
      cc.info[(int64_t)u * cc.ldi + i] = val;
    }
  }
}

// ------------------------------------------------------------------ backward
// Per unit (collapsed form):
//   dr_i  = K*ga_i + sum_j ga_{K+j} + p_i (dp_i - sum_k p_k dp_k),  dp_i = g1 . seq1_i
//   dz_i  = dr_i * [r_i > 0]
//   dseq1_i = p_i g1 + dz_i w1,   dseq2_i = g2/K + dz_i w2
//   dw1 += sum_i dz_i seq1_i,  dw2 += sum_i dz_i seq2_i,  dzsum = sum_i dz_i
__device__ __forceinline__ void atomic_add4(float* d, float4 v) {
  atomicAdd(d, v.x); atomicAdd(d + 1, v.y); atomicAdd(d + 2, v.z); atomicAdd(d + 3, v.w);
}

template <int KMAX, int SPL, bool ATOMIC>
__global__ __launch_bounds__(256, (KMAX <= 10 && SPL == 1) ? 4 : 1) void coattn_bwd_kernel_t(const CoattnArgs a) {
  extern __shared__ float lds[];  // [4 waves][upw][2*Dx]
  const int ci = (int)blockIdx.x >= a.c[1].first_block ? 1 : 0;
  const CoattnCall& cc = a.c[ci];
  const int GS = cc.GS, nslots = cc.nslots, F = cc.F, K = a.K, D4 = a.D4, n_units = a.n_units;
  const int mode = a.mode;
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int upw = 64 / GS;
  const int gl = lane & (GS - 1);
  const int grp = lane / GS;
  const int Dx = nslots * 4;
  const int D = D4 * 4;
  const int my_blocks = (ci == 0 ? min(a.c[1].first_block, (int)gridDim.x) : (int)gridDim.x) - cc.first_block;
  const int waves_total = my_blocks * 4;
  const int blk = (int)blockIdx.x - cc.first_block;
  const int wave0 = blk * 4 + wib;
  const float* __restrict__ table = a.table;
  float* __restrict__ gtable = a.gtable;
  const float* __restrict__ W = cc.W;

  bool sok[SPL];
  int f[SPL], coff[SPL];
  float4 w1[SPL], w2[SPL], dw1[SPL], dw2[SPL];
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    int s = gl + j * GS;
    sok[j] = s < nslots;
    int sc = sok[j] ? s : 0;
    f[j] = sc / D4;
    coff[j] = (sc - f[j] * D4) * 4;
    w1[j] = w2[j] = dw1[j] = dw2[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mode == 0) {
      w1[j] = ld4(W + Dx + sc * 4);
      w2[j] = ld4(W + 2 * Dx + sc * 4);
    }
  }

  const int n_iter = (n_units + waves_total * upw - 1) / (waves_total * upw);
  for (int it = 0; it < n_iter; ++it) {
    const int unit = (it * waves_total + wave0) * upw + grp;
    const bool unit_ok = unit < n_units;
    const int u = unit_ok ? unit : 0;
    const int ub = u / a.T;
    const int64_t ui = (int64_t)ub * a.Tidx + (u - ub * a.T);   // [B, Tidx, K, F] strides of the index tensors
    const int32_t* __restrict__ i1 = cc.idx1 + ui * K * F;
    const int32_t* __restrict__ i2 = cc.idx2 + ui * K * F;
    float4 g1[SPL], g2[SPL];
    bool ok[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
      ok[j] = unit_ok && sok[j];
      g1[j] = g2[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok[j]) {
        g1[j] = ld4(cc.g1 + (int64_t)u * cc.ld1 + (gl + j * GS) * 4);
        g2[j] = ld4(cc.g2 + (int64_t)u * cc.ld2 + (gl + j * GS) * 4);
      }
    }
    if (mode == 1) {  // RCA: d(sum_k row_k) = g for every k
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
          if (!ok[j] || k >= K) continue;
          if (!ATOMIC) continue;  // pull mode: every row gradient is g itself, nothing to prepare
          int64_t r1 = i1[k * F + f[j]], r2 = i2[k * F + f[j]];
          r1 = (uint64_t)r1 < a.n_rows ? r1 : 0;
          r2 = (uint64_t)r2 < a.n_rows ? r2 : 0;
          if (r1 != 0) atomic_add4(gtable + r1 * D + coff[j], g1[j]);
          if (r2 != 0) atomic_add4(gtable + r2 * D + coff[j], g2[j]);
        }
      }
      continue;
    }

    // pass 1: seq1 rows -> dp_k = g1 . seq1_k (seq1 stays in registers for dw1).  All 2K row ids, then the K
    // seq1 rows, are loaded unconditionally up front (clamped addresses, zeroed by selects afterwards): under
    // `if (k < K && ok)` every k was a region of its own that waited for its id and then for its row
    int32_t r1[SPL][KMAX], rb2[SPL][KMAX];
    float4 v1[SPL][KMAX];
    float dp[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int kc = k < K ? k : K - 1;
#pragma unroll
      for (int j = 0; j < SPL; ++j) {
        r1[j][k] = i1[kc * F + f[j]];
        rb2[j][k] = i2[kc * F + f[j]];
        r1[j][k] = (uint32_t)r1[j][k] < a.n_rows ? r1[j][k] : 0;      // (the forward reported it: score_state_t.id_status)
        rb2[j][k] = (uint32_t)rb2[j][k] < a.n_rows ? rb2[j][k] : 0;
      }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
#pragma unroll
      for (int j = 0; j < SPL; ++j) v1[j][k] = ld4(table + (int64_t)r1[j][k] * D + coff[j]);
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      dp[k] = 0.f;
#pragma unroll
      for (int j = 0; j < SPL; ++j) {
        const bool live = ok[j] && k < K;
        v1[j][k].x = live ? v1[j][k].x : 0.f; v1[j][k].y = live ? v1[j][k].y : 0.f;
        v1[j][k].z = live ? v1[j][k].z : 0.f; v1[j][k].w = live ? v1[j][k].w : 0.f;
        r1[j][k] = live ? r1[j][k] : 0;
        rb2[j][k] = live ? rb2[j][k] : 0;
      }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
#pragma unroll
      for (int j = 0; j < SPL; ++j) dp[k] += dot4(v1[j][k], g1[j]);
    // softmax from the saved relu'd scores
    float r[KMAX], p[KMAX], gik[KMAX];
    float rmax = 0.f, gsum = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {      // (clamped, unconditional loads: see above)
      const int kc = k < K ? k : K - 1;
      const float rv = cc.rsave[(int64_t)u * K + kc];
      const float g2v = cc.ginfo[(int64_t)u * cc.ldi + K + kc];
      gik[k] = cc.ginfo[(int64_t)u * cc.ldi + kc];
      r[k] = (k < K) ? rv : 0.f;
      rmax = fmaxf(rmax, r[k]);
      gsum += (k < K) ? g2v : 0.f;
    }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      p[k] = (k < K) ? expf(r[k] - rmax) : 0.f;
      den += p[k];
    }
    const float inv_den = 1.0f / den;
    float pdp = 0.f;
    group_sum_n<KMAX>(dp, GS);       // dp_k = g1 . seq1_k: the K dot products reduced across the group together
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      p[k] *= inv_den;
      dp[k] = (k < K) ? dp[k] : 0.f;
      pdp += p[k] * dp[k];
    }
    float dz[KMAX];
    float dzs = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      dz[k] = 0.f;
      if (k < K) {
        float dr = (float)K * gik[k] + gsum + p[k] * (dp[k] - pdp);
        dz[k] = r[k] > 0.f ? dr : 0.f;
        dzs += dz[k];
      }
    }
    if (unit_ok && gl == 0) cc.dzsum[u] = dzs;
    if (!ATOMIC && unit_ok) {  // the scalars the pull-form scatter multiplies G and w with
      for (int i = gl; i < K; i += GS) {
        float pv = 0.f, dv = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
          if (i == k) { pv = p[k]; dv = dz[k]; }
        cc.pcoef[(int64_t)u * K + i] = pv;
        cc.dzcoef[(int64_t)u * K + i] = dv;
      }
    }
    const float invK = 1.0f / (float)K;
    // dw1 += dz_k seq1_k, then the seq2 rows: dw2 += dz_k seq2_k.  The seq1 rows are READ AGAIN here (cache hits:
    // this wave fetched them a moment ago; masked uses carry row id 0, the all-zero row) instead of being held in
    // 40 VGPRs across the reductions and the softmax: 24 -> 10 spilled registers at 4 waves/SIMD, 0.262 -> 0.241 ms
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
      float4 xr[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) xr[k] = ld4(table + (int64_t)r1[j][k] * D + coff[j]);
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        dw1[j] = fma4(dz[k], xr[k], dw1[j]);
        if (ATOMIC && r1[j][k] != 0) {
          float4 d1 = fma4(dz[k], w1[j],
                           make_float4(p[k] * g1[j].x, p[k] * g1[j].y, p[k] * g1[j].z, p[k] * g1[j].w));
          atomic_add4(gtable + (int64_t)r1[j][k] * D + coff[j], d1);
        }
      }
    }
    // pass 2: the K seq2 rows, all in flight together (masked uses carry row id 0: the all-zero dummy row)
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
      const float4 g2k = make_float4(g2[j].x * invK, g2[j].y * invK, g2[j].z * invK, g2[j].w * invK);
      float4 y[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) y[k] = ld4(table + (int64_t)rb2[j][k] * D + coff[j]);
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        dw2[j] = fma4(dz[k], y[k], dw2[j]);
        if (ATOMIC && rb2[j][k] != 0) atomic_add4(gtable + (int64_t)rb2[j][k] * D + coff[j], fma4(dz[k], w2[j], g2k));
      }
    }
  }
  if (mode == 1) return;
  // block-level reduction of (dw1, dw2) -> slab[block][2*Dx], fixed order
  float* mine = lds + ((wib * upw + grp) * 2) * Dx;
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    if (!sok[j]) continue;
    st4(mine + (gl + j * GS) * 4, dw1[j]);
    st4(mine + Dx + (gl + j * GS) * 4, dw2[j]);
  }
  __syncthreads();
  const int nsrc = 4 * upw;
  for (int e = threadIdx.x; e < 2 * Dx; e += blockDim.x) {
    float s = 0.f;
    for (int q = 0; q < nsrc; ++q) s += lds[q * 2 * Dx + e];
    cc.slab[(int64_t)blk * 2 * Dx + e] = s;
  }
}

#define COATTN_DISPATCH(KERNEL, EXTRA, SPLV, KV, ...)                                     \
  do {                                                                                    \
    if (SPLV == 1) {                                                                      \
      if (KV <= 4) { hipLaunchKernelGGL((KERNEL<4, 1 EXTRA>), __VA_ARGS__); }             \
      else if (KV <= 10) { hipLaunchKernelGGL((KERNEL<10, 1 EXTRA>), __VA_ARGS__); }      \
      else if (KV <= 20) { hipLaunchKernelGGL((KERNEL<20, 1 EXTRA>), __VA_ARGS__); }      \
      else { hipLaunchKernelGGL((KERNEL<32, 1 EXTRA>), __VA_ARGS__); }                    \
    } else if (SPLV == 2) {                                                               \
      if (KV <= 4) { hipLaunchKernelGGL((KERNEL<4, 2 EXTRA>), __VA_ARGS__); }             \
      else if (KV <= 10) { hipLaunchKernelGGL((KERNEL<10, 2 EXTRA>), __VA_ARGS__); }      \
      else if (KV <= 20) { hipLaunchKernelGGL((KERNEL<20, 2 EXTRA>), __VA_ARGS__); }      \
      else { hipLaunchKernelGGL((KERNEL<32, 2 EXTRA>), __VA_ARGS__); }                    \
    } else {                                                                              \
      if (KV <= 10) { hipLaunchKernelGGL((KERNEL<10, 4 EXTRA>), __VA_ARGS__); }           \
      else { hipLaunchKernelGGL((KERNEL<20, 4 EXTRA>), __VA_ARGS__); }                    \
    }                                                                                     \
  } while (0)
#define COMMA_TRUE , true
#define COMMA_FALSE , false

static int coattn_check(const void* table, int64_t n_rows, int D, int F, int K, int B, int T) {
  if (!table || n_rows <= 0 || B <= 0 || T <= 0) return SCORE_E_BADARG;
  if (D <= 0 || (D & 3) || D > 256 || F <= 0 || K <= 0 || K > 32) return SCORE_E_SHAPE;
  if (F * (D / 4) > 256) return SCORE_E_SHAPE;
  if (F * (D / 4) > 128 && K > 20) return SCORE_E_SHAPE;
  return 0;
}

// Launch 1 or 2 calls (ncalls) of the forward in a single grid.
int score_coattn_fwd_multi(CoattnArgs& a, int ncalls, int D, int B, hipStream_t s) {
  int spl = 1, total = 0;
  a.D4 = D / 4;
  a.n_units = B * a.T;
  if (a.Tidx <= 0) a.Tidx = a.T;
  if (a.n_rows == 0) a.n_rows = 0x80000000u;      // (no row count given: int32 ids >= 0 pass)
  for (int c = 0; c < 2; ++c) {
    if (c >= ncalls) { a.c[c] = a.c[0]; a.c[c].first_block = 0x7fffffff; continue; }
    int SPLc;
    coattn_geom(D, a.c[c].F, &a.c[c].GS, &SPLc, &a.c[c].nslots);
    if (SPLc > spl) spl = SPLc;
    a.c[c].first_block = total;
    total += (int)cdiv64(cdiv64(a.n_units, 64 / a.c[c].GS), 4);
  }
  if (spl == 3) spl = 4;
  COATTN_DISPATCH(coattn_fwd_kernel, , spl, a.K, dim3(total), dim3(256), 0, s, a);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// Backward of 1 or 2 calls in a single grid + the per-call slab reductions into dW[c] (+=).
int score_coattn_bwd_multi(CoattnArgs& a, int ncalls, int D, int B, float* const dW[2], float* scratch,
                           int64_t scratch_floats, int atomic_scatter, ColsumJobs* cq, hipStream_t s) {
  int spl = 1, total = 0, nblk[2] = {0, 0};
  size_t lds_bytes = 0;
  int64_t used = 0;
  a.D4 = D / 4;
  a.n_units = B * a.T;
  if (a.Tidx <= 0) a.Tidx = a.T;
  if (a.n_rows == 0) a.n_rows = 0x80000000u;
  for (int c = 0; c < 2; ++c) {
    if (c >= ncalls) { a.c[c] = a.c[0]; a.c[c].first_block = 0x7fffffff; continue; }
    int SPLc;
    coattn_geom(D, a.c[c].F, &a.c[c].GS, &SPLc, &a.c[c].nslots);
    if (SPLc > spl) spl = SPLc;
    int upw = 64 / a.c[c].GS, Dx = a.c[c].nslots * 4;
    int blocks = (int)cdiv64(cdiv64(a.n_units, upw), 4);
    if (blocks > 512) blocks = 512;
    nblk[c] = blocks;
    a.c[c].first_block = total;
    total += blocks;
    a.c[c].slab = scratch + used;
    used += (int64_t)blocks * 2 * Dx;
    size_t l = (size_t)4 * upw * 2 * Dx * sizeof(float);
    if (l > lds_bytes) lds_bytes = l;
  }
  if (a.mode == 0 && used > scratch_floats) return SCORE_E_WORKSPACE;
  if (spl == 3) spl = 4;
  if (atomic_scatter) COATTN_DISPATCH(coattn_bwd_kernel_t, COMMA_TRUE, spl, a.K, dim3(total), dim3(256), lds_bytes, s, a);
  else COATTN_DISPATCH(coattn_bwd_kernel_t, COMMA_FALSE, spl, a.K, dim3(total), dim3(256), lds_bytes, s, a);
  SCORE_CHECK_LAUNCH();
  if (a.mode == 0) {
    for (int c = 0; c < ncalls; ++c) {
      int Dx = a.c[c].nslots * 4;
      if (cq) SCORE_TRY(colsum_queue_add(cq, a.c[c].slab, nblk[c], 2 * Dx, 2 * Dx, dW[c] + Dx, 1));
      else SCORE_TRY(score_launch_colsum(a.c[c].slab, nblk[c], 2 * Dx, 2 * Dx, dW[c] + Dx, 1, scratch + used,
                                         scratch_floats - used, s));
    }
  }
  return 0;
}

// ABI wrappers: one call; `tgt` is [B, F*D] contiguous
extern "C" int score_coattn_fwd(const float* table, int64_t n_rows, int32_t D, int32_t F, int32_t K,
                                int32_t B, int32_t T, const int32_t* idx1, const int32_t* idx2,
                                const float* tgt, const float* W, const float* bias, float* out1,
                                int32_t ld1, float* out2, int32_t ld2, float* info, int32_t ldi,
                                float* rsave, int32_t mode, void* stream) {
  SCORE_TRY(coattn_check(table, n_rows, D, F, K, B, T));
  if (!idx1 || !idx2 || !out1 || !out2) return SCORE_E_BADARG;
  if (mode == 0 && (!tgt || !W || !bias || !info || !rsave)) return SCORE_E_BADARG;
  if ((ld1 & 3) || (ld2 & 3)) return SCORE_E_SHAPE;
  CoattnArgs a;
  memset(&a, 0, sizeof(a));
  a.table = table; a.K = K; a.T = T; a.mode = mode; a.n_rows = (uint32_t)(n_rows < 0x80000000ll ? n_rows : 0x80000000ll);
  CoattnCall& c = a.c[0];
  c.idx1 = idx1; c.idx2 = idx2; c.tgt = tgt; c.ldt = F * D; c.W = W; c.bias = bias;
  c.out1 = out1; c.ld1 = ld1; c.out2 = out2; c.ld2 = ld2; c.info = info; c.ldi = ldi; c.rsave = rsave; c.F = F;
  return score_coattn_fwd_multi(a, 1, D, B, (hipStream_t)stream);
}

extern "C" int score_coattn_bwd(const float* table, float* grad_table, int64_t n_rows, int32_t D, int32_t F,
                                int32_t K, int32_t B, int32_t T, const int32_t* idx1, const int32_t* idx2,
                                const float* W, const float* rsave, const float* g1, int32_t ld1,
                                const float* g2, int32_t ld2, const float* ginfo, int32_t ldi, float* dzsum,
                                float* dW, float* scratch, int64_t scratch_floats, int32_t mode, void* stream) {
  SCORE_TRY(coattn_check(table, n_rows, D, F, K, B, T));
  if (!grad_table || !idx1 || !idx2 || !g1 || !g2) return SCORE_E_BADARG;
  if (mode == 0 && (!W || !rsave || !ginfo || !dzsum || !dW || !scratch)) return SCORE_E_BADARG;
  if ((ld1 & 3) || (ld2 & 3)) return SCORE_E_SHAPE;
  CoattnArgs a;
  memset(&a, 0, sizeof(a));
  a.table = table; a.gtable = grad_table; a.K = K; a.T = T; a.mode = mode;
  a.n_rows = (uint32_t)(n_rows < 0x80000000ll ? n_rows : 0x80000000ll);
  CoattnCall& c = a.c[0];
  c.idx1 = idx1; c.idx2 = idx2; c.W = W; c.rsave = const_cast<float*>(rsave); c.g1 = g1; c.ld1 = ld1; c.g2 = g2; c.ld2 = ld2;
  c.ginfo = ginfo; c.ldi = ldi; c.dzsum = dzsum; c.F = F;
  float* dWs[2] = {dW, nullptr};
  return score_coattn_bwd_multi(a, 1, D, B, dWs, scratch, scratch_floats, 1, nullptr, (hipStream_t)stream);
}

// ------------------------------------------------------------------ target rows (score.py:62-66, 210, 217)
__global__ void target_fwd_kernel(const float* __restrict__ table, int D4, int Fu, int Fi, int B,
                                  const int32_t* __restrict__ tu, const int32_t* __restrict__ ti,
                                  float* __restrict__ query, int ldq, float* __restrict__ head, int ldh,
                                  int off_ti, int off_tu, uint32_t n_rows, int32_t* __restrict__ id_status) {
  const int cu = Fu * D4, ci = Fi * D4;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * (cu + ci)) return;
  int b = (int)(i / (cu + ci));
  int s = (int)(i - (int64_t)b * (cu + ci));
  if (s < cu) {  // target_user slot
    int fidx = s / D4, c = s - fidx * D4;
    uint32_t row = (uint32_t)tu[b * Fu + fidx];
    if (row >= n_rows) {        // outside the table: the dummy row, reported (bit 4 = target_user)
      row = 0;
      if (id_status && c == 0) atomicOr(id_status, 1 << 4);
    }
    float4 v = ld4(table + ((int64_t)row * D4 + c) * 4);
    if (query) st4(query + (int64_t)b * ldq + s * 4, v);
    st4(head + (int64_t)b * ldh + off_tu + s * 4, v);
  } else {
    int s2 = s - cu;
    int fidx = s2 / D4, c = s2 - fidx * D4;
    uint32_t row = (uint32_t)ti[b * Fi + fidx];
    if (row >= n_rows) {        // (bit 5 = target_item)
      row = 0;
      if (id_status && c == 0) atomicOr(id_status, 1 << 5);
    }
    float4 v = ld4(table + ((int64_t)row * D4 + c) * 4);
    if (query) st4(query + (int64_t)b * ldq + cu * 4 + s2 * 4, v);
    st4(head + (int64_t)b * ldh + off_ti + s2 * 4, v);
  }
}

int score_launch_target_fwd(const float* table, int D, int Fu, int Fi, int B, const int32_t* tu,
                            const int32_t* ti, float* query, int ldq, float* head, int ldh, int off_ti,
                            int off_tu, hipStream_t s, int64_t n_rows, int32_t* id_status) {
  int64_t n = (int64_t)B * (Fu + Fi) * (D / 4);
  const uint32_t nr = (n_rows <= 0 || n_rows > 0x80000000ll) ? 0x80000000u : (uint32_t)n_rows;
  hipLaunchKernelGGL(target_fwd_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, table, D / 4, Fu, Fi, B,
                     tu, ti, query, ldq, head, ldh, off_ti, off_tu, nr, id_status);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// d target rows = dquery + dhead + S * w_t ; scatter-add into the table gradient.
// S[c][b] = sum_t dzsum_c[b*T+t] (c = 0: co-attention 1, whose target is the item; c = 1: the user) is summed here by
// every thread that needs it, in t order (the few rows of dzsum a block reads sit in L1), and stored once per (c, b)
// for the weight / bias gradients queued behind this launch -- it was a launch of its own in front of this one, two
// 5-us kernels on the critical path between the co-attention backward and the row scatter.
__global__ void target_bwd_kernel(float* __restrict__ gtable, int D4, int Fu, int Fi, int B, int T,
                                  const int32_t* __restrict__ tu, const int32_t* __restrict__ ti,
                                  const float* __restrict__ dquery, int ldq, const float* __restrict__ dhead,
                                  int ldh, int off_ti, int off_tu, const float* __restrict__ W1,
                                  const float* __restrict__ W2, const float* __restrict__ dz1,
                                  const float* __restrict__ dz2, float* __restrict__ S,
                                  float* __restrict__ dtgt, uint32_t n_rows) {
  const int cu = Fu * D4, ci = Fi * D4;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * (cu + ci)) return;
  int b = (int)(i / (cu + ci));
  int s = (int)(i - (int64_t)b * (cu + ci));
  float4 g;
  int64_t row;
  int c;
  const bool user = s < cu;
  float Sb = 0.f;
  if (S) {       // (co-attention models only)
    const float* src = user ? dz2 : dz1;
    if (src)
      for (int t = 0; t < T; ++t) Sb += src[(int64_t)b * T + t];
    if (s == 0) S[B + b] = Sb;
    if (s == cu) S[b] = Sb;
  }
  if (user) {
    int fidx = s / D4;
    c = s - fidx * D4;
    row = tu[b * Fu + fidx];
    g = ld4(dhead + (int64_t)b * ldh + off_tu + s * 4);
    if (dquery) g = add4(g, ld4(dquery + (int64_t)b * ldq + s * 4));
    if (W2) g = fma4(Sb, ld4(W2 + s * 4), g);  // co-attention 2 targets the user (score.py:197)
  } else {
    int s2 = s - cu;
    int fidx = s2 / D4;
    c = s2 - fidx * D4;
    row = ti[b * Fi + fidx];
    g = ld4(dhead + (int64_t)b * ldh + off_ti + s2 * 4);
    if (dquery) g = add4(g, ld4(dquery + (int64_t)b * ldq + cu * 4 + s2 * 4));
    if (W1) g = fma4(Sb, ld4(W1 + s2 * 4), g);     // co-attention 1 targets the item (score.py:196)
  }
  if (dtgt) {  // pull mode: hand the [B, Du+Di] row gradients to the sorted scatter
    st4(dtgt + (int64_t)b * (cu + ci) * 4 + s * 4, g);
  } else if (row != 0 && (uint64_t)row < n_rows) {      // (an id outside the table was read as the dummy row)
    atomic_add4(gtable + (row * D4 + c) * 4, g);
  }
}

int score_launch_target_bwd(float* grad_table, int D, int Fu, int Fi, int B, int T, const int32_t* tu,
                            const int32_t* ti, const float* dquery, int ldq, const float* dhead, int ldh,
                            int off_ti, int off_tu, const float* query, const float* W1, const float* W2,
                            const float* dzsum1, const float* dzsum2, float* S, float* dW1, float* dB1,
                            float* dW2, float* dB2, float* dtgt_out, float* scratch, int64_t scratch_floats,
                            ColsumJobs* cq, GemmQueue* gq, hipStream_t s, int64_t n_rows) {
  const bool coattn = W1 != nullptr;
  int64_t n = (int64_t)B * (Fu + Fi) * (D / 4);
  hipLaunchKernelGGL(target_bwd_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, grad_table, D / 4, Fu,
                     Fi, B, T, tu, ti, dquery, ldq, dhead, ldh, off_ti, off_tu, W1, W2, dzsum1, dzsum2,
                     coattn ? S : nullptr, dtgt_out,
                     (n_rows <= 0 || n_rows > 0x80000000ll) ? 0x80000000u : (uint32_t)n_rows);
  SCORE_CHECK_LAUNCH();
  if (coattn) {
    // dW_t = tgt^T S (call 0 targets the item: query cols Du.., call 1 the user: cols 0..), dbias = sum_b S
    const int Du = Fu * D, Di = Fi * D;
    if (cq) {   // one output column each: row-weighted column sums with the pass's other column sums (query and S stay
                // untouched until they run) -- as queued GEMM jobs they alone kept a launch of the f32 GEMM kernel in the end-of-pass
                // flush at cfg-3
      SCORE_TRY(colsum_queue_add(cq, query + Du, B, Di, ldq, dW1, 0, S));
      SCORE_TRY(colsum_queue_add(cq, query, B, Du, ldq, dW2, 0, S + B));
    } else if (gq) {
      SCORE_TRY(gemm_queue_add(gq, Di, 1, B, query + Du, ldq, S, 1, dW1, 1));
      SCORE_TRY(gemm_queue_add(gq, Du, 1, B, query, ldq, S + B, 1, dW2, 1));
    } else {
      SCORE_TRY(score_gemm(2, Di, 1, B, query + Du, ldq, S, 1, dW1, 1, nullptr, 0, 1.f, nullptr, 0, scratch,
                           scratch_floats, s));
      SCORE_TRY(score_gemm(2, Du, 1, B, query, ldq, S + B, 1, dW2, 1, nullptr, 0, 1.f, nullptr, 0, scratch,
                           scratch_floats, s));
    }
    if (cq) {
      SCORE_TRY(colsum_queue_add(cq, S, B, 1, 1, dB1, 0));
      SCORE_TRY(colsum_queue_add(cq, S + B, B, 1, 1, dB2, 0));
    } else {
      SCORE_TRY(score_launch_colsum(S, B, 1, 1, dB1, 0, scratch, scratch_floats, s));
      SCORE_TRY(score_launch_colsum(S + B, B, 1, 1, dB2, 0, scratch, scratch_floats, s));
    }
  }
  return 0;
}
