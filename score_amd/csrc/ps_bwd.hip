// Whole-model backward, one workgroup per sample (persample.h): from dz2 (the forward kernel writes it) down to every
// operand of the pass's weight-gradient products / column sums and of the sorted row scatter -- the backward of
// build_fc_net (score.py:68-76), the temporal attention (:169-186, 210-215), both recurrences (:205-208), their input
// projections and both co-attentions (:147-167, 196-201) in ONE launch.
#include <string.h>
#include "ps_device.h"
#include "kernels.h"

namespace {
#if defined(PS_PHASE_TIMING)
__device__ unsigned long long ps_ts_bwd[32];
#endif

// co-attention backward of one (slice, call) unit by a group of GS lanes: the body of coattn_bwd_kernel_t (embed.hip),
// pull form (no row gradient is written: the scatter gets the per-(unit, i) scalars p_i, dz_i)
//   dr_i = K*ga_i + sum_j ga_{K+j} + p_i (dp_i - sum_k p_k dp_k),  dp_i = g1 . seq1_i,  dz_i = dr_i [r_i > 0]
//   dw1 += sum_i dz_i seq1_i,  dw2 += sum_i dz_i seq2_i
template <int KMAX, int c>
__device__ __forceinline__ void ps_coattn_bwd(const PsBwdArgs& a, const PsLds& L, float* sm, int b, int v, float4& dw1,
                                              float4& dw2) {
  const PsShape& s = a.s;
  const int GS = PS2(s.GS, c), nslots = PS2(s.nslots, c), K = s.K, D4 = s.D4, D = 4 * D4, A = s.A, I = s.I;
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
  const int32_t* __restrict__ i1 = PS2(a.idx1, c) + ui * K * F;
  const int32_t* __restrict__ i2 = PS2(a.idx2, c) + ui * K * F;
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
  const int col1 = c == 0 ? 0 : s.Di;
  const float4 g1 = ps_sel4(ok, *reinterpret_cast<const float4*>(sm + L.b_dxs + (0 * A + tc) * I + col1 + sl * 4));
  float dp[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const bool live = ok && k < K;
    v1[k] = ps_sel4(live, v1[k]);
    yv[k] = ps_sel4(live, yv[k]);
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
      PS2(a.pcoef, c)[row * K + i] = pv;
      PS2(a.dzcoef, c)[row * K + i] = dv;
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
  PS_MARK(ps_ts_bwd, 0);
  if (b == 0 && a.gw && tid < a.npad)
    for (int e = 0; e < a.pad_len[tid]; ++e) a.gw[a.pad_off[tid] + e] = 0.f;

  // the backward images (contiguous: fc2t .. wxt) on their way into this XCD's L2 meanwhile
  const PsTouch warm = ps_touch(a.img + a.im.fc2t, a.im.total - a.im.fc2t, tid);
  // ---- phase 0: what the forward pass saved for this sample -> LDS.  Every array by its own wave(s): a wave then waits for
  // one or two round trips, all eight at once (as eleven loops of the whole workgroup each loop waited for its own loads)
  if (wave == 0) {
    for (int e = lane; e < 80; e += 64) sm[L.b_dz2 + e] = a.dz2[(int64_t)b * 80 + e];
    for (int e = lane; e < 208; e += 64) {
      sm[L.b_f1 + e] = e < 200 ? a.f1[(int64_t)b * 200 + e] : 0.f;
      if (e >= 200) sm[L.b_dz1 + e] = 0.f;
    }
    for (int e = lane; e < Dh; e += 64) sm[L.b_x + e] = a.head_inp[(int64_t)b * Dh + e];
  } else if (wave == 1) {
    for (int e = lane; e < A; e += 64) sm[L.b_sc + e] = a.att_score[bt0 + e];
    for (int e = lane; e < 2 * A * H; e += 64) {
      const int side = e / (A * H), r = e - side * A * H;
      sm[L.b_gout + e] = PS2(a.gru_out, side)[bt0 * H + r];
    }
  } else if (wave == 2) {
    for (int e = lane; e < A * 48; e += 64) {
      const int t = e / 48, n = e - t * 48;
      sm[L.b_a2 + t * L.ld2 + n] = n < 40 ? a.a2[(bt0 + t) * 40 + n] : 0.f;
    }
  } else if (wave < 5) {
    for (int e = (wave - 3) * 64 + lane; e < A * 20; e += 128) {        // float4 pieces of the saved a1 rows
      const int t = e / 20, n = (e - t * 20) * 4;
      *reinterpret_cast<float4*>(sm + L.b_a1 + t * L.ld1 + n) = ld4(a.a1 + (bt0 + t) * 80 + n);
    }
  } else if (wave < 7) {
    const int Dk4 = Dk >> 2;
    for (int e = (wave - 5) * 64 + lane; e < A * Dk4; e += 128) {        // the k half of the saved [k, q*k] rows
      const int t = e / Dk4, j = (e - t * Dk4) * 4;
      *reinterpret_cast<float4*>(sm + L.b_kk + t * Dk + j) = ld4(a.ainp + (bt0 + t) * 2 * Dk + j);
    }
  } else {
    for (int e = lane; e < ps_up(Dk, 16); e += 64) {
      sm[L.b_qv + e] = e < Dk ? a.q[(int64_t)b * Dk + e] : 0.f;
      if (e >= Dk) sm[L.b_dqv + e] = 0.f;
    }
    for (int e = lane; e < 2 * A * K; e += 64) {
      const int c = e / (A * K), r = e - c * A * K;
      sm[L.b_rs + e] = PS2(a.rsave, c)[bt0 * K + r];
    }
  }
  __syncthreads();
  PS_MARK(ps_ts_bwd, 1);

  // ---- phase 1: dz1 = [f1 > 0] (dz2 . W2^T) / keep   (13 column tiles: two per wave where needed)
  {
    float go[2];
    const float4* const tl[2] = {ps_tile(a.img, a.im.fc2t, wave, 5, 13), ps_tile(a.img, a.im.fc2t, wave + 8, 5, 13)};
    ps_gemv<2>(go, sm + L.b_dz2, tl, 5, lane);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int col = (wave + 8 * i) * 16 + lc;
      if (lq == 0 && col < 200) {
        const float v = sm[L.b_f1 + col] > 0.f ? go[i] / a.keep : 0.f;
        sm[L.b_dz1 + col] = v;
        a.dz1[(int64_t)b * 200 + col] = v;
      }
    }
  }
  __syncthreads();
  PS_MARK(ps_ts_bwd, 2);
  // ---- phase 2: d bn1 = dz1 . W1^T; d head = d bn * gamma * rs; bn1's d gamma terms d bn * x * rs
  {
    const int nt = (Dh + 15) >> 4;
    float go[2];
    const float4* const tl[2] = {ps_tile(a.img, a.im.fc1t, wave, 13, nt), ps_tile(a.img, a.im.fc1t, wave + 8, 13, nt)};
    ps_gemv<2>(go, sm + L.b_dz1, tl, 13, lane);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int col = (wave + 8 * i) * 16 + lc;
      if (lq == 0 && col < Dh) {
        const float d = go[i];
        a.dbn[(int64_t)b * Dh + col] = d;
        a.dgstage[(int64_t)b * Dh + col] = d * (sm[L.b_x + col] * a.rs);
        sm[L.b_dh + col] = d * (W[a.bn_g + col] * a.rs);
      }
    }
  }
  __syncthreads();
  PS_MARK(ps_ts_bwd, 3);
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
  PS_MARK(ps_ts_bwd, 4);
  for (int e = tid; e < A * 40; e += PS_NT) {      // da2[t][n] = ds_t * w5[n] * [a2 > 0]  (in place)
    const int t = e / 40, n = e - t * 40;
    const float dv = sm[L.b_a2 + t * L.ld2 + n] > 0.f ? sm[L.b_dsv + t] * W[a.at_w5 + n] : 0.f;
    sm[L.b_a2 + t * L.ld2 + n] = dv;
    a.da2[(bt0 + t) * 40 + n] = dv;
  }
  __syncthreads();
  PS_MARK(ps_ts_bwd, 5);
  // ---- phase 4: da1 = [a1 > 0] (da2 . W4^T)  (in place over the saved a1)
  {
    ps_f32x4 acc[1][MT];
    ps_zero<MT, 1>(acc);
    const float4* const tl[1] = {ps_tile(a.img, a.im.w4t, wave, 3, 5)};
    ps_mma<MT, 1>(acc, sm + L.b_a2, L.ld2, tl, 3, lane);
    const int col = wave * 16 + lc;
    if (wave < 5) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = m * 16 + 4 * lq + v;
          if (row < A) {
            const float d = sm[L.b_a1 + row * L.ld1 + col] > 0.f ? acc[0][m][v] : 0.f;
            sm[L.b_a1 + row * L.ld1 + col] = d;
            a.da1[(bt0 + row) * 80 + col] = d;
          }
        }
    }
  }
  __syncthreads();
  PS_MARK(ps_ts_bwd, 6);
  // sum_t da1 -> the gradient reaching the per-sample q term of the folded dense_3
  for (int n = tid; n < 80; n += PS_NT) {
    float acc = 0.f;
    for (int t = 0; t < A; ++t) acc += sm[L.b_a1 + t * L.ld1 + n];
    sm[L.b_adz + n] = acc;
    a.adzsum[(int64_t)b * 80 + n] = acc;
  }
  __syncthreads();
  PS_MARK(ps_ts_bwd, 7);
  // ---- phase 5: d inp = da1 . Weff^T ([A, 2 Dk]: two column tiles per wave) and the q term's dqd = adzsum . (Wa + Wc)^T
  {
    const int ntd = (2 * Dk + 15) >> 4, nqd = (Dk + 15) >> 4;
    ps_f32x4 acc[2][MT];
    ps_zero<MT, 2>(acc);
    const float4* const tl[2] = {ps_tile(a.img, a.im.wefft, wave, 5, ntd), ps_tile(a.img, a.im.wefft, wave + 8, 5, ntd)};
    float acq[1];
    const float4* const tq[1] = {ps_tile(a.img, a.im.wqt, wave, 5, nqd)};
    ps_mma<MT, 2>(acc, sm + L.b_a1, L.ld1, tl, 5, lane);
    ps_gemv<1>(acq, sm + L.b_adz, tq, 5, lane);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int col = (wave + 8 * i) * 16 + lc;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = m * 16 + 4 * lq + v;
          if (row < A && col < 2 * Dk) sm[L.b_dainp + row * L.lda + col] = acc[i][m][v];
        }
    }
    const int colq = wave * 16 + lc;
    if (lq == 0 && colq < Dk) sm[L.b_dqd + colq] = acq[0];
  }
  __syncthreads();
  PS_MARK(ps_ts_bwd, 8);
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
  PS_MARK(ps_ts_bwd, 9);

  // ---- phase 7: the two backward recurrences, one wave per side (a lane owns column j = lane % 32: row j of the recurrent kernels
  // in registers; the vectors a step multiplies them with reach every lane as scalars, v_readlane, as in the forward kernel);
  // waves 2-7 meanwhile: d query = dq . W^T (dense_2)
  //   dh = dout_t + dh_next; du = dh (h_prev - c); dc = dh (1 - u); dh_prev = dh u
  //   dpc = dc (1 - c^2); d(rh) = dpc . Wc^T; dr = d(rh) h_prev; dh_prev += d(rh) r
  //   dpr = dr r (1 - r); dpu = du u (1 - u); dh_prev += [dpr, dpu] . Wg^T
  if (wave < 2) {
    const int side = wave, j = lane & 31;
    const float* __restrict__ Wg = W + PS2(a.gk, side) + (int64_t)I * 2 * H + (int64_t)j * 2 * H;      // row j of the h rows
    const float* __restrict__ Wcn = W + PS2(a.ck, side) + (int64_t)I * H + (int64_t)j * H;
    float wgr[H], wgu[H], wct[H];
#pragma unroll
    for (int k4 = 0; k4 < H / 4; ++k4) {
      const float4 x = ld4(Wg + 4 * k4), y = ld4(Wg + H + 4 * k4), z = ld4(Wcn + 4 * k4);
      wgr[4 * k4] = x.x; wgr[4 * k4 + 1] = x.y; wgr[4 * k4 + 2] = x.z; wgr[4 * k4 + 3] = x.w;
      wgu[4 * k4] = y.x; wgu[4 * k4 + 1] = y.y; wgu[4 * k4 + 2] = y.z; wgu[4 * k4 + 3] = y.w;
      wct[4 * k4] = z.x; wct[4 * k4 + 1] = z.y; wct[4 * k4 + 2] = z.z; wct[4 * k4 + 3] = z.w;
    }
    const float* gsave = PS2(a.gates, side) + bt0 * 3 * H;
    const float* gob = sm + L.b_gout + side * A * H;
    const float* dgb = sm + L.b_dgru + side * A * H;
    float* dxl = sm + L.b_dxp + side * MP * L.lddx;
    float* dxg = PS2(a.dxproj, side) + bt0 * 3 * H;
    float* rhg = PS2(a.rh, side) + bt0 * H;
    float* hpg = PS2(a.hprev, side) + bt0 * H;
    float dh = 0.f;
    float nr = gsave[(A - 1) * 3 * H + j], nu = gsave[(A - 1) * 3 * H + H + j], nc = gsave[(A - 1) * 3 * H + 2 * H + j];
    float nhp = A > 1 ? gob[(A - 2) * H + j] : 0.f, ndo = dgb[(A - 1) * H + j];
    for (int t = A - 1; t >= 0; --t) {
      const float r = nr, u = nu, cnd = nc, hp = nhp, dout = ndo;
      const int tp = t > 0 ? t - 1 : 0;          // the next step's saved gates / state / gradient, requested a step ahead
      nr = gsave[tp * 3 * H + j]; nu = gsave[tp * 3 * H + H + j]; nc = gsave[tp * 3 * H + 2 * H + j];
      nhp = t > 1 ? gob[(t - 2) * H + j] : 0.f;
      ndo = dgb[tp * H + j];
      const bool live = t < len;
      const float d = dh + dout;
      const float du = d * (hp - cnd), dc = d * (1.0f - u);
      const float dpu = live ? du * u * (1.0f - u) : 0.f;
      const float dpc = live ? dc * (1.0f - cnd * cnd) : 0.f;
      dh = live ? d * u : dh;
      float p0 = 0.f, p1 = 0.f;
#pragma unroll
      for (int k = 0; k < H; k += 2) {
        const float x0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dpc), k));
        const float x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dpc), k + 1));
        p0 = fmaf(x0, wct[k], p0);
        p1 = fmaf(x1, wct[k + 1], p1);
      }
      const float drh = p0 + p1;
      const float dpr = live ? drh * hp * r * (1.0f - r) : 0.f;
      dh = live ? fmaf(drh, r, dh) : dh;
      float q0 = 0.f, q1 = 0.f;
#pragma unroll
      for (int k = 0; k < H; ++k) {
        const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dpr), k));
        const float y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dpu), k));
        q0 = fmaf(x, wgr[k], q0);
        q1 = fmaf(y, wgu[k], q1);
      }
      dh += q0 + q1;                              // (zero past the length: dpr = dpu = 0 there)
      if (lane < H) {
        dxl[t * L.lddx + j] = dpr; dxl[t * L.lddx + H + j] = dpu; dxl[t * L.lddx + 2 * H + j] = dpc;
        dxg[t * 3 * H + j] = dpr; dxg[t * 3 * H + H + j] = dpu; dxg[t * 3 * H + 2 * H + j] = dpc;
        rhg[t * H + j] = live ? r * hp : 0.f;
        hpg[t * H + j] = live ? hp : 0.f;
      }
    }
  } else {
    const int nti = (I + 15) >> 4, nck = (Dk + 15) >> 4;
    float go[2];
    const float4* const tl[2] = {ps_tile(a.img, a.im.q2t, wave - 2, nck, nti), ps_tile(a.img, a.im.q2t, wave + 4, nck, nti)};
    ps_gemv<2>(go, sm + L.b_dqv, tl, nck, lane);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int col = (wave - 2 + 6 * i) * 16 + lc;
      if (lq == 0 && col < I) sm[L.b_dquery + col] = go[i];
    }
  }
  __syncthreads();
  PS_MARK(ps_ts_bwd, 10);
  // ---- phase 8: d x = [dgates | dcand] . [Wx_gates | Wx_cand]^T for both sides (waves 0-3 / 4-7: up to three tiles each)
  {
    const int nti = (I + 15) >> 4;
    const int side = wave >> 2, cw = wave & 3;
    ps_f32x4 acc[3][MT];
    ps_zero<MT, 3>(acc);
    const int64_t io = PS2(a.im.wxt, side);
    const float4* const tl[3] = {ps_tile(a.img, io, cw, 6, nti), ps_tile(a.img, io, cw + 4, 6, nti), ps_tile(a.img, io, cw + 8, 6, nti)};
    ps_mma<MT, 3>(acc, sm + L.b_dxp + side * MP * L.lddx, L.lddx, tl, 6, lane);
    float* dxo = PS2(a.dxside, side);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int col = (cw + 4 * i) * 16 + lc;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = m * 16 + 4 * lq + v;
          if (row < A && col < I) {
            sm[L.b_dxs + (side * A + row) * I + col] = acc[i][m][v];
            dxo[(bt0 + row) * I + col] = acc[i][m][v];
          }
        }
    }
  }
  __syncthreads();
  PS_MARK(ps_ts_bwd, 11);
  // ---- phase 9: both co-attentions' backward; per-group partial sums of dW1 | dW2 into the slabs (dxp is dead)
  {
    float4 dw1[2], dw2[2];
    dw1[0] = dw1[1] = dw2[0] = dw2[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int v0 = wave * 64; v0 < s.Vtot; v0 += PS_NT) {
      if (v0 >= s.V0) ps_coattn_bwd<KMAX, 1>(a, L, sm, b, v0 + lane, dw1[1], dw2[1]);
      else ps_coattn_bwd<KMAX, 0>(a, L, sm, b, v0 + lane, dw1[0], dw2[0]);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int GS = PS2(s.GS, c), Dx = PS2(s.nslots, c) * 4;
      const int gl = lane & (GS - 1), grp = wave * (64 / GS) + lane / GS;
      if (gl < PS2(s.nslots, c)) {
        float* mine = sm + L.b_slab + c * 8 * PS_NT + grp * 2 * Dx;
        *reinterpret_cast<float4*>(mine + gl * 4) = dw1[c];
        *reinterpret_cast<float4*>(mine + Dx + gl * 4) = dw2[c];
      }
    }
  }
  __syncthreads();
  PS_MARK(ps_ts_bwd, 12);
  // ---- phase 10: the sample's dW1 | dW2 (fixed order over the groups); S_c = sum_t dzsum_c; d target rows
  for (int c = 0; c < 2; ++c) {
    const int Dx = PS2(s.nslots, c) * 4, ng = PS_NT / PS2(s.GS, c);
    for (int e = tid; e < 2 * Dx; e += PS_NT) {
      float acc = 0.f;
      for (int q = 0; q < ng; ++q) acc += sm[L.b_slab + c * 8 * PS_NT + q * 2 * Dx + e];
      PS2(a.caslab, c)[(int64_t)b * 2 * Dx + e] = acc;
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
      g = fma4(Sb, ld4(W + PS2(a.ca_w, c) + s2 * 4), g);
      st4(a.dtgt + (int64_t)b * I + sl * 4, g);
      if (sl == 0) a.S[s.B + b] = Sb;
      if (sl == cu) a.S[b] = Sb;
    }
  }
  ps_touch_use(warm, sm + L.b_dzs);
  PS_MARK(ps_ts_bwd, 13);
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

#if defined(PS_PHASE_TIMING)
extern "C" int score_ps_phase_read_bwd(unsigned long long* out32) {
  return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(ps_ts_bwd), 32 * sizeof(unsigned long long));
}
#endif
