// Whole-model backward, one workgroup per sample (persample.h): from dz2 (the forward kernel writes it) down to every
// operand of the pass's weight-gradient products / column sums and of the sorted row scatter -- the backward of
// build_fc_net (score.py:68-76), the temporal attention (:169-186, 210-215), both recurrences (:205-208), their input
// projections and both co-attentions (:147-167, 196-201) in ONE launch.
#include <string.h>
#include "ps_device.h"
#include "kernels.h"

namespace {

// co-attention backward of one (slice, call) unit by a group of GS lanes: the body of coattn_bwd_kernel_t (embed.hip),
// pull form (no row gradient is written: the scatter gets the per-(unit, i) scalars p_i, dz_i)
//   dr_i = K*ga_i + sum_j ga_{K+j} + p_i (dp_i - sum_k p_k dp_k),  dp_i = g1 . seq1_i,  dz_i = dr_i [r_i > 0]
//   dw1 += sum_i dz_i seq1_i,  dw2 += sum_i dz_i seq2_i
template <int KMAX>
__device__ __forceinline__ void ps_coattn_bwd(const PsBwdArgs& a, const PsLds& L, float* sm, int b, int v, int c, float4& dw1,
                                              float4& dw2) {
  const PsShape& s = a.s;
  const int GS = s.GS[c], nslots = s.nslots[c], K = s.K, D4 = s.D4, D = 4 * D4, A = s.A, I = s.I;
  const int F = c == 0 ? s.Fi : s.Fu;
  const int rel = v - (c ? s.V0 : 0);
  const int t = rel / GS, gl = rel & (GS - 1);
  const bool unit_ok = t < A;
  const int tc = unit_ok ? t : 0;
  const bool ok = unit_ok && gl < nslots;
  const int sl = gl < nslots ? gl : 0;
  const int f = sl / D4, coff = (sl - f * D4) * 4;
  const float* __restrict__ table = a.table;
  const int64_t ui = (int64_t)b * s.Tidx + tc;
  const int32_t* __restrict__ i1 = a.idx1[c] + ui * K * F;
  const int32_t* __restrict__ i2 = a.idx2[c] + ui * K * F;
  int32_t ra[KMAX], rb[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int kc = k < K ? k : K - 1;
    ra[k] = i1[kc * F + f];
    rb[k] = i2[kc * F + f];
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {       // (the forward pass reported ids outside the table: score_state_t.id_status)
    ra[k] = (uint32_t)ra[k] < a.n_rows ? ra[k] : 0;
    rb[k] = (uint32_t)rb[k] < a.n_rows ? rb[k] : 0;
  }
  float4 v1[KMAX], yv[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    v1[k] = ld4(table + (int64_t)ra[k] * D + coff);
    yv[k] = ld4(table + (int64_t)rb[k] * D + coff);
  }
  const int col1 = c == 0 ? 0 : s.Di, col2 = c == 0 ? s.Du : 0;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 g1 = ok ? *reinterpret_cast<const float4*>(sm + L.b_dxs + (0 * A + tc) * I + col1 + sl * 4) : z4;
  float dp[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const bool live = ok && k < K;
    v1[k] = live ? v1[k] : z4;
    yv[k] = live ? yv[k] : z4;
    dp[k] = dot4(v1[k], g1);
  }
  float r[KMAX], p[KMAX], gik[KMAX];
  float rmax = 0.f, gsum = 0.f;
  const float* rsv = sm + L.b_rs + (c * A + tc) * K;
  const float* gi = sm + L.b_dinfo + tc * 4 * K + c * 2 * K;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int kc = k < K ? k : K - 1;
    r[k] = (k < K) ? rsv[kc] : 0.f;
    gik[k] = gi[kc];
    gsum += (k < K) ? gi[K + kc] : 0.f;
    rmax = fmaxf(rmax, r[k]);
  }
  float den = 0.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    p[k] = (k < K) ? expf(r[k] - rmax) : 0.f;
    den += p[k];
  }
  const float inv_den = 1.0f / den;
  group_sum_n<KMAX>(dp, GS);
  float pdp = 0.f;
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
      const float dr = (float)K * gik[k] + gsum + p[k] * (dp[k] - pdp);
      dz[k] = r[k] > 0.f ? dr : 0.f;
      dzs += dz[k];
    }
  }
  if (unit_ok) {
    if (gl == 0) sm[L.b_dzs + c * s.MP + t] = dzs;
    const int64_t row = (int64_t)b * A + t;
    for (int i = gl; i < K; i += GS) {
      float pv = 0.f, dv = 0.f;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (i == k) { pv = p[k]; dv = dz[k]; }
      a.pcoef[c][row * K + i] = pv;
      a.dzcoef[c][row * K + i] = dv;
    }
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const float d = unit_ok ? dz[k] : 0.f;
    dw1 = fma4(d, v1[k], dw1);
    dw2 = fma4(d, yv[k], dw2);
  }
}

template <int KMAX, int MT>
__global__ __launch_bounds__(PS_NT) void ps_bwd_kernel(const PsBwdArgs a) {
  extern __shared__ float sm[];
  constexpr int H = 32;
  const PsShape& s = a.s;
  PsLds L;
  ps_lds_layout(s, &L);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, lq = lane >> 4;
  const int A = s.A, MP = s.MP, I = s.I, Dk = s.Dk, Dh = s.Dhead, K = s.K;
  const float* __restrict__ W = a.W;
  const int len = min(a.length[b], A);
  const int64_t bt0 = (int64_t)b * A;

  // ---- phase 0: what the forward pass saved for this sample -> LDS (every load independent of the others)
  for (int e = tid; e < 80; e += PS_NT) sm[L.b_dz2 + e] = a.dz2[(int64_t)b * 80 + e];
  for (int e = tid; e < 208; e += PS_NT) {
    sm[L.b_f1 + e] = e < 200 ? a.f1[(int64_t)b * 200 + e] : 0.f;
    if (e >= 200) sm[L.b_dz1 + e] = 0.f;
  }
  for (int e = tid; e < Dh; e += PS_NT) sm[L.b_x + e] = a.head_inp[(int64_t)b * Dh + e];
  for (int e = tid; e < A; e += PS_NT) sm[L.b_sc + e] = a.att_score[bt0 + e];
  for (int e = tid; e < 2 * A * H; e += PS_NT) {
    const int side = e / (A * H), r = e - side * A * H;
    sm[L.b_gout + e] = a.gru_out[side][bt0 * H + r];
  }
  for (int e = tid; e < A * 48; e += PS_NT) {
    const int t = e / 48, n = e - t * 48;
    sm[L.b_a2 + t * L.ld2 + n] = n < 40 ? a.a2[(bt0 + t) * 40 + n] : 0.f;
  }
  for (int e = tid; e < A * 80; e += PS_NT) {
    const int t = e / 80, n = e - t * 80;
    sm[L.b_a1 + t * L.ld1 + n] = a.a1[(bt0 + t) * 80 + n];
  }
  for (int e = tid; e < A * Dk; e += PS_NT) {
    const int t = e / Dk, j = e - t * Dk;
    sm[L.b_kk + e] = a.ainp[(bt0 + t) * 2 * Dk + j];
  }
  for (int e = tid; e < ps_up(Dk, 16); e += PS_NT) {
    sm[L.b_qv + e] = e < Dk ? a.q[(int64_t)b * Dk + e] : 0.f;
    if (e >= Dk) sm[L.b_dqv + e] = 0.f;
  }
  for (int e = tid; e < 2 * A * K; e += PS_NT) {
    const int c = e / (A * K), r = e - c * A * K;
    sm[L.b_rs + e] = a.rsave[c][bt0 * K + r];
  }
  __syncthreads();

  // ---- phase 1: dz1 = [f1 > 0] (dz2 . W2^T) / keep
  for (int ct = wave; ct < 13; ct += PS_NW) {
    ps_f32x4 acc[1];
    ps_zero<1>(acc);
    ps_mma<1>(acc, sm + L.b_dz2, 0, ps_tile(a.img, a.im.fc2t, ct, 5), 5, lane);
    const int col = ct * 16 + lc;
    if (lq == 0 && col < 200) {
      const float v = sm[L.b_f1 + col] > 0.f ? acc[0][0] / a.keep : 0.f;
      sm[L.b_dz1 + col] = v;
      a.dz1[(int64_t)b * 200 + col] = v;
    }
  }
  __syncthreads();
  // ---- phase 2: d bn1 = dz1 . W1^T; d head = d bn * gamma * rs; bn1's d gamma terms d bn * x * rs
  {
    const int nt = (Dh + 15) >> 4;
    for (int ct = wave; ct < nt; ct += PS_NW) {
      ps_f32x4 acc[1];
      ps_zero<1>(acc);
      ps_mma<1>(acc, sm + L.b_dz1, 0, ps_tile(a.img, a.im.fc1t, ct, 13), 13, lane);
      const int col = ct * 16 + lc;
      if (lq == 0 && col < Dh) {
        const float d = acc[0][0];
        a.dbn[(int64_t)b * Dh + col] = d;
        a.dgstage[(int64_t)b * Dh + col] = d * (sm[L.b_x + col] * a.rs);
        sm[L.b_dh + col] = d * (W[a.bn_g + col] * a.rs);
      }
    }
  }
  __syncthreads();
  // ---- phase 3: pooling + masked softmax + dense_5 backward; a lane per slice
  //   dscore_t = duf . ur_t + dif . ir_t ; ds_t = score_t (dscore_t - sum score * dscore) [t < len]
  if (wave == 0) {
    const bool tok = lane < A;
    const int t = tok ? lane : 0;
    float part = 0.f;
    if (s.off_u >= 0)
      for (int j = 0; j < H; ++j) part = fmaf(sm[L.b_dh + s.off_u + j], sm[L.b_gout + (0 * A + t) * H + j], part);
    if (s.off_i >= 0)
      for (int j = 0; j < H; ++j) part = fmaf(sm[L.b_dh + s.off_i + j], sm[L.b_gout + (1 * A + t) * H + j], part);
    const float scv = tok ? sm[L.b_sc + t] : 0.f;
    const float tot = wave_sum(tok ? scv * part : 0.f);
    const float g = (tok && lane < len) ? scv * (part - tot) : 0.f;
    if (tok) {
      sm[L.b_dsv + t] = g;
      a.ds[bt0 + t] = g;
    }
  }
  __syncthreads();
  for (int e = tid; e < A * 40; e += PS_NT) {      // da2[t][n] = ds_t * w5[n] * [a2 > 0]  (in place)
    const int t = e / 40, n = e - t * 40;
    const float dv = sm[L.b_a2 + t * L.ld2 + n] > 0.f ? sm[L.b_dsv + t] * W[a.at_w5 + n] : 0.f;
    sm[L.b_a2 + t * L.ld2 + n] = dv;
    a.da2[(bt0 + t) * 40 + n] = dv;
  }
  __syncthreads();
  // ---- phase 4: da1 = [a1 > 0] (da2 . W4^T)  (in place over the saved a1)
  for (int ct = wave; ct < 5; ct += PS_NW) {
    ps_f32x4 acc[MT];
    ps_zero<MT>(acc);
    ps_mma<MT>(acc, sm + L.b_a2, L.ld2, ps_tile(a.img, a.im.w4t, ct, 3), 3, lane);
    const int col = ct * 16 + lc;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = m * 16 + 4 * lq + v;
        if (row < A) {
          const float d = sm[L.b_a1 + row * L.ld1 + col] > 0.f ? acc[m][v] : 0.f;
          sm[L.b_a1 + row * L.ld1 + col] = d;
          a.da1[(bt0 + row) * 80 + col] = d;
        }
      }
  }
  __syncthreads();
  // sum_t da1 -> the gradient reaching the per-sample q term of the folded dense_3
  for (int n = tid; n < 80; n += PS_NT) {
    float acc = 0.f;
    for (int t = 0; t < A; ++t) acc += sm[L.b_a1 + t * L.ld1 + n];
    sm[L.b_adz + n] = acc;
    a.adzsum[(int64_t)b * 80 + n] = acc;
  }
  __syncthreads();
  // ---- phase 5: d inp = da1 . Weff^T ([A, 2 Dk]) and the q term's dqd = adzsum . (Wa + Wc)^T
  {
    const int ntd = (2 * Dk + 15) >> 4, nqd = (Dk + 15) >> 4;
    for (int task = wave; task < ntd + nqd; task += PS_NW) {
      if (task < ntd) {
        ps_f32x4 acc[MT];
        ps_zero<MT>(acc);
        ps_mma<MT>(acc, sm + L.b_a1, L.ld1, ps_tile(a.img, a.im.wefft, task, 5), 5, lane);
        const int col = task * 16 + lc;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int row = m * 16 + 4 * lq + v;
            if (row < A && col < 2 * Dk) sm[L.b_dainp + row * L.lda + col] = acc[m][v];
          }
      } else {
        const int ct = task - ntd;
        ps_f32x4 acc[1];
        ps_zero<1>(acc);
        ps_mma<1>(acc, sm + L.b_adz, 0, ps_tile(a.img, a.im.wqt, ct, 5), 5, lane);
        const int col = ct * 16 + lc;
        if (lq == 0 && col < Dk) sm[L.b_dqd + col] = acc[0][0];
      }
    }
  }
  __syncthreads();
  // ---- phase 6: backward of [k, q*k] and of the pooled-state path: d states, d atten_info, dq  (a thread per column)
  for (int j = tid; j < Dk; j += PS_NT) {
    const float qq = sm[L.b_qv + j];
    float pooled = 0.f;
    if (j < H && s.off_u >= 0) pooled = sm[L.b_dh + s.off_u + j];
    if (j >= H && j < 2 * H && s.off_i >= 0) pooled = sm[L.b_dh + s.off_i + (j - H)];
    float dqa = 0.f;
    for (int t = 0; t < A; ++t) {
      const float d1 = sm[L.b_dainp + t * L.lda + j], d3 = sm[L.b_dainp + t * L.lda + Dk + j];
      const float kv = sm[L.b_kk + t * Dk + j];
      dqa = fmaf(d3, kv, dqa);
      const float dk = fmaf(d3, qq, d1) + pooled * sm[L.b_sc + t];
      if (j < H) sm[L.b_dgru + (0 * A + t) * H + j] = dk;
      else if (j < 2 * H) sm[L.b_dgru + (1 * A + t) * H + (j - H)] = dk;
      else sm[L.b_dinfo + t * 4 * K + (j - 2 * H)] = dk;
    }
    const float dq = dqa + sm[L.b_dqd + j];
    sm[L.b_dqv + j] = dq;
    a.dq[(int64_t)b * Dk + j] = dq;
  }
  __syncthreads();

  // ---- phase 7: both backward recurrences on wave 0 (a lane owns column j of its side: row j of the recurrent kernels in
  // registers); the other waves: d query = dq . W^T (dense_2)
  //   dh = dout_t + dh_next; du = dh (h_prev - c); dc = dh (1 - u); dh_prev = dh u
  //   dpc = dc (1 - c^2); d(rh) = dpc . Wc^T; dr = d(rh) h_prev; dh_prev += d(rh) r
  //   dpr = dr r (1 - r); dpu = du u (1 - u); dh_prev += [dpr, dpu] . Wg^T
  if (wave == 0) {
    const int side = lane >> 5, j = lane & 31;
    const float* __restrict__ Wg = W + a.gk[side] + (int64_t)I * 2 * H + (int64_t)j * 2 * H;      // row j of the h rows
    const float* __restrict__ Wcn = W + a.ck[side] + (int64_t)I * H + (int64_t)j * H;
    float wgr[H], wgu[H], wct[H];
#pragma unroll
    for (int k4 = 0; k4 < H / 4; ++k4) {
      const float4 x = ld4(Wg + 4 * k4), y = ld4(Wg + H + 4 * k4), z = ld4(Wcn + 4 * k4);
      wgr[4 * k4] = x.x; wgr[4 * k4 + 1] = x.y; wgr[4 * k4 + 2] = x.z; wgr[4 * k4 + 3] = x.w;
      wgu[4 * k4] = y.x; wgu[4 * k4 + 1] = y.y; wgu[4 * k4 + 2] = y.z; wgu[4 * k4 + 3] = y.w;
      wct[4 * k4] = z.x; wct[4 * k4 + 1] = z.y; wct[4 * k4 + 2] = z.z; wct[4 * k4 + 3] = z.w;
    }
    float* dpcs = sm + L.b_hs + side * H;
    float* dprs = sm + L.b_hs + 2 * H + side * H;
    float* dpus = sm + L.b_hs + 4 * H + side * H;
    const float* gsave = a.gates[side] + bt0 * 3 * H;
    const float* gob = sm + L.b_gout + side * A * H;
    const float* dgb = sm + L.b_dgru + side * A * H;
    float* dxl = sm + L.b_dxp + side * MP * L.lddx;
    float* dxg = a.dxproj[side] + bt0 * 3 * H;
    float* rhg = a.rh[side] + bt0 * H;
    float* hpg = a.hprev[side] + bt0 * H;
    float dh = 0.f;
    float nr = gsave[(A - 1) * 3 * H + j], nu = gsave[(A - 1) * 3 * H + H + j], nc = gsave[(A - 1) * 3 * H + 2 * H + j];
    for (int t = A - 1; t >= 0; --t) {
      const float r = nr, u = nu, cnd = nc;
      const int tp = t > 0 ? t - 1 : 0;          // the next step's saved gates, requested a step ahead
      nr = gsave[tp * 3 * H + j]; nu = gsave[tp * 3 * H + H + j]; nc = gsave[tp * 3 * H + 2 * H + j];
      const bool live = t < len;
      const float hp = t > 0 ? gob[(t - 1) * H + j] : 0.f;
      const float d = dh + dgb[t * H + j];
      const float du = d * (hp - cnd), dc = d * (1.0f - u);
      const float dpu = live ? du * u * (1.0f - u) : 0.f;
      const float dpc = live ? dc * (1.0f - cnd * cnd) : 0.f;
      dh = live ? d * u : dh;
      dpcs[j] = dpc;
      ps_wave_sync();
      float drh = 0.f;
#pragma unroll
      for (int k4 = 0; k4 < H / 4; ++k4) {
        const float4 x = *reinterpret_cast<const float4*>(dpcs + 4 * k4);
        drh = fmaf(x.x, wct[4 * k4], drh); drh = fmaf(x.y, wct[4 * k4 + 1], drh);
        drh = fmaf(x.z, wct[4 * k4 + 2], drh); drh = fmaf(x.w, wct[4 * k4 + 3], drh);
      }
      const float dpr = live ? drh * hp * r * (1.0f - r) : 0.f;
      dh = live ? fmaf(drh, r, dh) : dh;
      dprs[j] = dpr;
      dpus[j] = dpu;
      ps_wave_sync();
      float acc = 0.f;
#pragma unroll
      for (int k4 = 0; k4 < H / 4; ++k4) {
        const float4 x = *reinterpret_cast<const float4*>(dprs + 4 * k4);
        const float4 y = *reinterpret_cast<const float4*>(dpus + 4 * k4);
        acc = fmaf(x.x, wgr[4 * k4], acc); acc = fmaf(x.y, wgr[4 * k4 + 1], acc);
        acc = fmaf(x.z, wgr[4 * k4 + 2], acc); acc = fmaf(x.w, wgr[4 * k4 + 3], acc);
        acc = fmaf(y.x, wgu[4 * k4], acc); acc = fmaf(y.y, wgu[4 * k4 + 1], acc);
        acc = fmaf(y.z, wgu[4 * k4 + 2], acc); acc = fmaf(y.w, wgu[4 * k4 + 3], acc);
      }
      dh += acc;                                  // (zero past the length: dpr = dpu = 0 there)
      dxl[t * L.lddx + j] = dpr; dxl[t * L.lddx + H + j] = dpu; dxl[t * L.lddx + 2 * H + j] = dpc;
      dxg[t * 3 * H + j] = dpr; dxg[t * 3 * H + H + j] = dpu; dxg[t * 3 * H + 2 * H + j] = dpc;
      rhg[t * H + j] = live ? r * hp : 0.f;
      hpg[t * H + j] = live ? hp : 0.f;
      ps_wave_sync();
    }
  } else {
    const int nti = (I + 15) >> 4, nck = (Dk + 15) >> 4;
    for (int ct = wave - 1; ct < nti; ct += PS_NW - 1) {
      ps_f32x4 acc[1];
      ps_zero<1>(acc);
      ps_mma<1>(acc, sm + L.b_dqv, 0, ps_tile(a.img, a.im.q2t, ct, nck), nck, lane);
      const int col = ct * 16 + lc;
      if (lq == 0 && col < I) sm[L.b_dquery + col] = acc[0][0];
    }
  }
  __syncthreads();
  // ---- phase 8: d x = [dgates | dcand] . [Wx_gates | Wx_cand]^T for both sides
  {
    const int nti = (I + 15) >> 4;
    for (int task = wave; task < 2 * nti; task += PS_NW) {
      const int side = task / nti, ct = task - side * nti;
      ps_f32x4 acc[MT];
      ps_zero<MT>(acc);
      ps_mma<MT>(acc, sm + L.b_dxp + side * MP * L.lddx, L.lddx, ps_tile(a.img, a.im.wxt[side], ct, 6), 6, lane);
      const int col = ct * 16 + lc;
      // (the tile's results are parked in registers until every wave is done reading dxp: dxs does not overlap it, but
      //  the dW slabs written two phases on do -- nothing to wait for here)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = m * 16 + 4 * lq + v;
          if (row < A && col < I) {
            sm[L.b_dxs + (side * A + row) * I + col] = acc[m][v];
            a.dxside[side][(bt0 + row) * I + col] = acc[m][v];
          }
        }
    }
  }
  __syncthreads();
  // ---- phase 9: both co-attentions' backward; per-group partial sums of dW1 | dW2 into the slabs (dxp is dead)
  {
    float4 dw1[2], dw2[2];
    dw1[0] = dw1[1] = dw2[0] = dw2[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int v0 = wave * 64; v0 < s.Vtot; v0 += PS_NT) {
      if (v0 >= s.V0) ps_coattn_bwd<KMAX>(a, L, sm, b, v0 + lane, 1, dw1[1], dw2[1]);
      else ps_coattn_bwd<KMAX>(a, L, sm, b, v0 + lane, 0, dw1[0], dw2[0]);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int GS = s.GS[c], Dx = s.nslots[c] * 4;
      const int gl = lane & (GS - 1), grp = wave * (64 / GS) + lane / GS;
      if (gl < s.nslots[c]) {
        float* mine = sm + L.b_slab + c * 8 * PS_NT + grp * 2 * Dx;
        *reinterpret_cast<float4*>(mine + gl * 4) = dw1[c];
        *reinterpret_cast<float4*>(mine + Dx + gl * 4) = dw2[c];
      }
    }
  }
  __syncthreads();
  // ---- phase 10: the sample's dW1 | dW2 (fixed order over the groups); S_c = sum_t dzsum_c; d target rows
  for (int c = 0; c < 2; ++c) {
    const int Dx = s.nslots[c] * 4, ng = PS_NT / s.GS[c];
    for (int e = tid; e < 2 * Dx; e += PS_NT) {
      float acc = 0.f;
      for (int q = 0; q < ng; ++q) acc += sm[L.b_slab + c * 8 * PS_NT + q * 2 * Dx + e];
      a.caslab[c][(int64_t)b * 2 * Dx + e] = acc;
    }
  }
  {
    // d target rows = d query + d head (target columns) + S * w_t  (call 0 targets the item, call 1 the user: score.py:196-197)
    const int cu = s.Fu * s.D4, nq4 = cu + s.Fi * s.D4;
    for (int sl = tid; sl < nq4; sl += PS_NT) {
      const bool user = sl < cu;
      const int c = user ? 1 : 0;
      float Sb = 0.f;
      for (int t = 0; t < A; ++t) Sb += sm[L.b_dzs + c * MP + t];
      const int s2 = user ? sl : sl - cu;
      float4 g = *reinterpret_cast<const float4*>(sm + L.b_dh + (user ? s.off_tu : s.off_ti) + s2 * 4);
      g = add4(g, *reinterpret_cast<const float4*>(sm + L.b_dquery + sl * 4));
      g = fma4(Sb, ld4(W + a.ca_w[c] + s2 * 4), g);
      st4(a.dtgt + (int64_t)b * I + sl * 4, g);
      if (sl == 0) a.S[s.B + b] = Sb;
      if (sl == cu) a.S[b] = Sb;
    }
  }
}

}  // namespace

template <int KMAX, int MT>
static int ps_bwd_launch(const PsBwdArgs& a, size_t lds, hipStream_t s) {
  if (lds > 48 * 1024) {
    static bool done[SCORE_PS_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SCORE_PS_MAX_DEVICES) return SCORE_E_BADARG;
    if (!done[dev]) {
      hipError_t e = hipFuncSetAttribute((const void*)ps_bwd_kernel<KMAX, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
      done[dev] = true;
    }
  }
  hipLaunchKernelGGL((ps_bwd_kernel<KMAX, MT>), dim3(a.s.B), dim3(PS_NT), lds, s, a);
  SCORE_CHECK_LAUNCH();
  return 0;
}

int score_launch_ps_bwd(const PsBwdArgs& a, hipStream_t s) {
  PsLds L;
  ps_lds_layout(a.s, &L);
  const size_t lds = (size_t)L.bwd_total * 4;
  const int mt = a.s.MP / 16;
  if (a.s.K <= 5) {
    if (mt == 1) return ps_bwd_launch<5, 1>(a, lds, s);
    if (mt == 2) return ps_bwd_launch<5, 2>(a, lds, s);
    return ps_bwd_launch<5, 3>(a, lds, s);
  }
  if (mt == 1) return ps_bwd_launch<10, 1>(a, lds, s);
  if (mt == 2) return ps_bwd_launch<10, 2>(a, lds, s);
  return ps_bwd_launch<10, 3>(a, lds, s);
}
