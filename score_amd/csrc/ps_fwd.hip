// Whole-model forward, one workgroup per sample (persample.h): score.py:188-224 + build_fc_net / build_logloss
// (:68-81) in ONE launch for the reference's own shapes, and the per-step weight images it runs on.
#include <string.h>
#include "ps_device.h"
#include "kernels.h"

namespace {
#if defined(PS_PHASE_TIMING)
__device__ unsigned long long ps_ts_fwd[32];
#endif

// ------------------------------------------------------------------------------------------------ weight images
__device__ __forceinline__ float ps_src(const PsImgJob& j, int r, int c) {
  switch (j.kind) {
    case PS_SRC_WXCAT: { const int H = j.aux; return c < 2 * H ? j.W[(int64_t)r * 2 * H + c] : j.W2[(int64_t)r * H + (c - 2 * H)]; }
    case PS_SRC_WEFF: {
      const int Dk = j.aux;
      return r < Dk ? j.W[(int64_t)(Dk + r) * j.ld + c] - j.W[(int64_t)(2 * Dk + r) * j.ld + c]
                    : j.W[(int64_t)(2 * Dk + r) * j.ld + c];
    }
    case PS_SRC_WQ: { const int Dk = j.aux; return j.W[(int64_t)r * j.ld + c] + j.W[(int64_t)(2 * Dk + r) * j.ld + c]; }
    default: return j.W[(int64_t)r * j.ld + c];
  }
}

#define PS_L2_PARTS 256
__global__ __launch_bounds__(256) void ps_prep_kernel(const PsPrepArgs a) {
  const int blk = blockIdx.x;
  if (blk < a.img_blocks) {
    int ji = 0;
    for (int q = 1; q < a.njobs; ++q)
      if (blk >= a.job[q].first_block) ji = q;
    const PsImgJob& j = a.job[ji];
    const int nchunk = (j.K + 15) >> 4, nct = (j.N + 15) >> 4;
    const int e = (blk - j.first_block) * 256 + (int)threadIdx.x;
    if (e >= nct * nchunk * 64) return;
    const int lane = e & 63, c = (e >> 6) % nchunk, ct = (e >> 6) / nchunk;
    const int n = ct * 16 + (lane & 15), k0 = c * 16 + 4 * (lane >> 4);
    float v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = k0 + s;
      v[s] = (k < j.K && n < j.N) ? (j.trans ? ps_src(j, n, k) : ps_src(j, k, n)) : 0.f;
    }
    st4(j.img + (int64_t)e * 4, make_float4(v[0], v[1], v[2], v[3]));
    return;
  }
  const int b2 = blk - a.img_blocks;
  if (b2 < PS_L2_PARTS) {       // partial sums of squares of the regularised range (build_l2norm, score.py:91-94)
    __shared__ float sh[256];
    const int64_t n4 = a.n_reg >> 2;
    float s = 0.f;
    for (int64_t i = (int64_t)b2 * 256 + threadIdx.x; i < n4; i += (int64_t)PS_L2_PARTS * 256) {
      const float4 v = ld4(a.wreg + i * 4);
      s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
    }
    if (b2 == 0 && threadIdx.x < (unsigned)(a.n_reg - n4 * 4)) { const float v = a.wreg[n4 * 4 + threadIdx.x]; s = fmaf(v, v, s); }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      a.part[b2] = sh[0];
      if (b2 == 0) reinterpret_cast<unsigned int*>(a.part)[PS_L2_PARTS] = 0u;       // the forward kernel's count of finished workgroups
    }
    return;
  }
  // the dense gradient starts from zero (the backward pass accumulates some of its pieces)
  const int b3 = b2 - PS_L2_PARTS;
  const int64_t i = ((int64_t)b3 * 256 + threadIdx.x) * 4;
  if (i + 3 < a.zero_floats) st4(a.zero + i, make_float4(0.f, 0.f, 0.f, 0.f));
  else for (int64_t e = i; e < a.zero_floats; ++e) a.zero[e] = 0.f;
}

// ------------------------------------------------------------------------------------------------ forward
// one (slice, co-attention call) unit by a group of GS lanes: the body of coattn_fwd_kernel (embed.hip) with one slot per
// lane, the target rows gathered by the group itself, results into LDS and global memory
template <int KMAX, int c>
__device__ __forceinline__ void ps_gather(const PsFwdArgs& a, const PsLds& L, float* sm, int b, int v) {
  const PsShape& s = a.s;
  const int GS = PS2(s.GS, c), nslots = PS2(s.nslots, c), K = s.K, D4 = s.D4, D = 4 * D4, A = s.A;
  const int F = c == 0 ? s.Fi : s.Fu;
  const int rel = v - (c ? s.V0 : 0);
  const int t = rel / GS, gl = rel & (GS - 1);
  const bool unit_ok = t < A;
  const int tc = unit_ok ? t : 0;
  const bool ok = unit_ok && gl < nslots;
  const int sl = gl < nslots ? gl : 0;
  const int f = sl / D4, coff = (sl - f * D4) * 4;
  const int Dx = nslots * 4;
  const float* __restrict__ table = a.table;
  const float* __restrict__ Wc = a.W + PS2(a.ca_w, c);
  const float4 wt = ld4(Wc + sl * 4), w1 = ld4(Wc + Dx + sl * 4), w2 = ld4(Wc + 2 * Dx + sl * 4);
  const int64_t ui = (int64_t)b * s.Tidx + tc;
  const int32_t* __restrict__ i1 = PS2(a.idx1, c) + ui * K * F;
  const int32_t* __restrict__ i2 = PS2(a.idx2, c) + ui * K * F;
  const int32_t* __restrict__ tg = (c == 0 ? a.ti : a.tu) + (int64_t)b * F;
  // every load of the unit is unconditional and goes out before anything is consumed (embed.hip: 2K dependent round
  // trips per wave otherwise): the 2K + 1 row ids, then the 2K + 1 rows
  int32_t ra[KMAX], rb[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int kc = k < K ? k : K - 1;
    ra[k] = i1[kc * F + f];
    rb[k] = i2[kc * F + f];
  }
  int32_t rt = tg[f];
  const uint32_t NR = a.n_rows;
  bool bad1 = false, bad2 = false;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const bool b1 = (uint32_t)ra[k] >= NR, b2 = (uint32_t)rb[k] >= NR;
    bad1 |= b1; bad2 |= b2;
    ra[k] = b1 ? 0 : ra[k];
    rb[k] = b2 ? 0 : rb[k];
  }
  rt = (uint32_t)rt >= NR ? 0 : rt;          // (reported by the wave that writes the target rows)
  // bits = position of the tensor in the feed tuple (graph_loader.py:383): call 0 reads user_1hop (0) / item_2hop (3),
  // call 1 user_2hop (1) / item_1hop (2)
  if (a.id_status && (bad1 || bad2))
    atomicOr(a.id_status, (bad1 ? (c == 0 ? 1 : 2) : 0) | (bad2 ? (c == 0 ? 8 : 4) : 0));
  float4 v1[KMAX], yv[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    v1[k] = ld4(table + (int64_t)ra[k] * D + coff);
    yv[k] = ld4(table + (int64_t)rb[k] * D + coff);
  }
  const float4 tv = ld4(table + (int64_t)rt * D + coff);
  float4 sum2 = make_float4(0.f, 0.f, 0.f, 0.f);
  float red[KMAX + 1];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const bool live = ok && k < K;
    v1[k] = ps_sel4(live, v1[k]);
    const float4 y = ps_sel4(live, yv[k]);
    sum2 = add4(sum2, y);
    red[k] = (k < K) ? dot4(v1[k], w1) + dot4(y, w2) : 0.f;
  }
  red[KMAX] = ok ? dot4(tv, wt) : 0.f;
  group_sum_n<KMAX + 1>(red, GS);
  const float cc = red[KMAX] + a.W[PS2(a.ca_b, c)];
  float r[KMAX], p[KMAX];
  float rmax = 0.f, rsum = 0.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    r[k] = 0.f;
    if (k < K) {
      r[k] = fmaxf(red[k] + cc, 0.f);
      rmax = fmaxf(rmax, r[k]);
      rsum += r[k];
    }
  }
  float den = 0.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    p[k] = (k < K) ? expf(r[k] - rmax) : 0.f;
    den += p[k];
  }
  const float inv_den = 1.0f / den;
  // user_side = [user_1hop_seq | user_2hop_seq], item_side = [item_1hop_seq | item_2hop_seq]  (score.py:196-201):
  // call 0 = (user_1hop, item_2hop, target_item), call 1 = (user_2hop, item_1hop, target_user)
  const int col1 = c == 0 ? 0 : s.Di, col2 = c == 0 ? s.Du : 0;
  if (ok) {
    float4 o = z4;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) o = fma4(p[k] * inv_den, v1[k], o);
    const float fk = (float)K;
    const float4 o2 = make_float4(sum2.x / fk, sum2.y / fk, sum2.z / fk, sum2.w / fk);
    const int64_t row = (int64_t)b * A + t;
    st4(a.xside[0] + row * s.I + col1 + gl * 4, o);
    st4(a.xside[1] + row * s.I + col2 + gl * 4, o2);
    *reinterpret_cast<float4*>(sm + L.xs + (0 * s.MP + t) * L.ldx + col1 + gl * 4) = o;
    *reinterpret_cast<float4*>(sm + L.xs + (1 * s.MP + t) * L.ldx + col2 + gl * 4) = o2;
  }
  // atten_info = [K*r_0..K*r_{K-1}, sum_i r_i (K times)]  (score.py:165-166); [info_item | info_user] (:198)
  if (unit_ok) {
    const int64_t row = (int64_t)b * A + t;
    for (int i = gl; i < 2 * K; i += GS) {
      float val = rsum, rv = 0.f;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (i == k) rv = r[k];
      if (i < K) {
        val = (float)K * rv;
        PS2(a.rsave, c)[row * K + i] = rv;
      }
      a.info[row * 4 * K + c * 2 * K + i] = val;
      sm[L.infos + t * 4 * K + c * 2 * K + i] = val;
    }
  }
}

__device__ __forceinline__ float ps_act(float v, float bias, int drop, float keep, const uint8_t* mask, uint64_t seed, int row,
                                        int col, int N) {
  v = fmaxf(v + bias, 0.f);                             // dense(activation=relu)
  if (drop) {                                           // tf.nn.dropout: x / keep * Bernoulli(keep); element numbering of
    const uint64_t e = (uint64_t)row * (uint64_t)N + (uint64_t)col;      // head_fused.hip / the GEMM epilogue
    const bool on = mask ? (mask[e] != 0) : (hash_uniform(seed, e) < keep);
    v = on ? v / keep : 0.f;
  }
  return v;
}

template <int KMAX, int MT>
__global__ __launch_bounds__(PS_NT) void ps_fwd_kernel(const PsFwdArgs a) {
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
  const uint64_t seed0 = a.seed_dev ? *a.seed_dev : a.seed0;
  const uint64_t seed1 = a.seed_dev ? (seed0 ^ 0x5DEECE66Dull) : a.seed1;
  PS_MARK(ps_ts_fwd, 0);

  // the forward images (contiguous: wx .. fc2) on their way into this XCD's L2 while the gather runs
  const PsTouch warm = ps_touch(a.img + a.im.wx[0], a.im.fc2t - a.im.wx[0], tid);
  // ---- phase 1: target rows (last wave) and the fused gather + both co-attentions (everybody)
  {
    // zero the k padding of the operands built in this phase
    const int Ip = ps_up(I, 16);
    for (int e = tid; e < 2 * MP * (Ip - I); e += PS_NT) {
      const int r = e / (Ip - I), cidx = e - r * (Ip - I);
      sm[L.xs + r * L.ldx + I + cidx] = 0.f;
    }
    for (int e = I + tid; e < Ip; e += PS_NT) sm[L.qs + e] = 0.f;
    for (int e = Dk + tid; e < ps_up(Dk, 16); e += PS_NT) sm[L.qv + e] = 0.f;
    for (int e = Dh + tid; e < ps_up(Dh, 16); e += PS_NT) sm[L.bns + e] = 0.f;
    for (int e = 200 + tid; e < 208; e += PS_NT) sm[L.f1s + e] = 0.f;
  }
  if (wave == PS_NW - 1) {
    // query = [target_user | target_item] (score.py:210), head_inp = [.., target_item, target_user] (:217)
    const int cu = s.Fu * s.D4, nq4 = cu + s.Fi * s.D4;
    for (int sl = lane; sl < nq4; sl += 64) {
      const bool user = sl < cu;
      const int s2 = user ? sl : sl - cu;
      const int f = s2 / s.D4, cidx = s2 - f * s.D4;
      uint32_t row = (uint32_t)(user ? a.tu[(int64_t)b * s.Fu + f] : a.ti[(int64_t)b * s.Fi + f]);
      if (row >= a.n_rows) {
        row = 0;
        if (a.id_status && cidx == 0) atomicOr(a.id_status, user ? 1 << 4 : 1 << 5);
      }
      const float4 v = ld4(a.table + ((int64_t)row * s.D4 + cidx) * 4);
      st4(a.query + (int64_t)b * I + sl * 4, v);
      const int hoff = (user ? s.off_tu : s.off_ti) + s2 * 4;
      st4(a.head_inp + (int64_t)b * Dh + hoff, v);
      *reinterpret_cast<float4*>(sm + L.qs + sl * 4) = v;
      *reinterpret_cast<float4*>(sm + L.hin + hoff) = v;
    }
  }
  for (int v0 = wave * 64; v0 < s.Vtot; v0 += PS_NT) {
    if (v0 >= s.V0) ps_gather<KMAX, 1>(a, L, sm, b, v0 + lane);
    else ps_gather<KMAX, 0>(a, L, sm, b, v0 + lane);
  }
  __syncthreads();
  PS_MARK(ps_ts_fwd, 1);

  // ---- phase 2: GRU input projections of both sides (x . [Wx_gates | Wx_cand] + bias: waves 0-5, two column tiles each) and
  // the attention's query q (waves 6-7, four tiles each): every tile of the phase in flight at once
  {
    const int ncx = (I + 15) >> 4, nq = (Dk + 15) >> 4;
    if (wave < 6) {
      const int side = wave / 3, ct0 = 2 * (wave % 3);
      ps_f32x4 acc[2][MT];
      ps_zero<MT, 2>(acc);
      const int64_t io = PS2(a.im.wx, side);
      const float4* const tl[2] = {ps_tile(a.img, io, ct0, ncx), ps_tile(a.img, io, ct0 + 1, ncx)};
      ps_mma<MT, 2>(acc, sm + L.xs + side * MP * L.ldx, L.ldx, tl, ncx, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int col = (ct0 + i) * 16 + lc;
        const float bias = col < 2 * H ? W[PS2(a.gb, side) + col] : W[PS2(a.cb, side) + col - 2 * H];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int row = m * 16 + 4 * lq + v;
            if (row < A) sm[L.xp + (side * A + row) * 3 * H + col] = acc[i][m][v] + bias;
          }
      }
    } else {
      const int ctb = (wave - 6) * 4;
      float qo[4];
      const float4* const tl[4] = {ps_tile(a.img, a.im.q2, ctb, ncx, nq), ps_tile(a.img, a.im.q2, ctb + 1, ncx, nq),
                                   ps_tile(a.img, a.im.q2, ctb + 2, ncx, nq), ps_tile(a.img, a.im.q2, ctb + 3, ncx, nq)};
      ps_gemv<4>(qo, sm + L.qs, tl, ncx, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = (ctb + i) * 16 + lc;
        if (lq == 0 && col < Dk) {
          const float v = qo[i] + W[a.at_b[0] + col];
          sm[L.qv + col] = v;
          a.q[(int64_t)b * Dk + col] = v;
        }
      }
    }
  }
  __syncthreads();
  PS_MARK(ps_ts_fwd, 2);

  // ---- phase 3: the two recurrences, one wave per side (wave 0 user side, wave 1 item side).  A lane owns column j = lane % 32 of
  // r, u and the candidate with its 3 x 32 recurrent weights in registers (both half-waves compute the same thing); the state
  // h_k reaches every lane as a scalar, v_readlane from lane k -- no LDS round trip in the step (through LDS, 8 + 8
  // ds_read_b128 per step and their latency were the step: 0.6 us).  Waves 2-6 meanwhile: the per-sample term of the folded
  // dense_3, qz = q . (Wa + Wc) + b
  if (wave < 2) {
    const int side = wave, j = lane & 31;
    const float* __restrict__ Wg = W + PS2(a.gk, side) + (int64_t)I * 2 * H;      // h rows of gates/kernel [H, 2H]
    const float* __restrict__ Wcn = W + PS2(a.ck, side) + (int64_t)I * H;          // h rows of candidate/kernel [H, H]
    float wr[H], wu[H], wc[H];
#pragma unroll
    for (int k = 0; k < H; ++k) {
      wr[k] = Wg[k * 2 * H + j];
      wu[k] = Wg[k * 2 * H + H + j];
      wc[k] = Wcn[k * H + j];
    }
    const float* xpb = sm + L.xp + side * A * 3 * H;
    float* gob = sm + L.gout + side * MP * H;
    float* gsave = PS2(a.gates, side) + (int64_t)b * A * 3 * H;
    float* osave = PS2(a.gru_out, side) + (int64_t)b * A * H;
    float h = 0.f;
    float nxr = xpb[j], nxu = xpb[H + j], nxc = xpb[2 * H + j];
    for (int t = 0; t < A; ++t) {
      const float xr = nxr, xu = nxu, xc = nxc;
      const int tn = t + 1 < A ? t + 1 : t;            // the next step's x-projection, read a step ahead
      nxr = xpb[tn * 3 * H + j]; nxu = xpb[tn * 3 * H + H + j]; nxc = xpb[tn * 3 * H + 2 * H + j];
      float ar0 = xr, ar1 = 0.f, au0 = xu, au1 = 0.f;
#pragma unroll
      for (int k = 0; k < H; k += 2) {
        const float h0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h), k));
        const float h1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h), k + 1));
        ar0 = fmaf(h0, wr[k], ar0); au0 = fmaf(h0, wu[k], au0);
        ar1 = fmaf(h1, wr[k + 1], ar1); au1 = fmaf(h1, wu[k + 1], au1);
      }
      const float r = ps_sigmoid(ar0 + ar1), u = ps_sigmoid(au0 + au1);
      const float rh = r * h;
      float ac0 = xc, ac1 = 0.f;
#pragma unroll
      for (int k = 0; k < H; k += 2) {
        const float g0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rh), k));
        const float g1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rh), k + 1));
        ac0 = fmaf(g0, wc[k], ac0);
        ac1 = fmaf(g1, wc[k + 1], ac1);
      }
      const float cnd = ps_tanh(ac0 + ac1);
      const float hn = u * h + (1.0f - u) * cnd;
      const bool live = t < len;
      const float o = live ? hn : 0.f;         // dynamic_rnn: zero output past the length, state carried through
      h = live ? hn : h;
      if (lane < H) {
        gsave[t * 3 * H + j] = r; gsave[t * 3 * H + H + j] = u; gsave[t * 3 * H + 2 * H + j] = cnd;
        osave[t * H + j] = o;
        gob[t * H + j] = o;
      }
    }
    if (lane < H && PS2(a.gru_final, side)) PS2(a.gru_final, side)[(int64_t)b * H + j] = h;
  } else {
    const int nck = (Dk + 15) >> 4;
    float qo[1];
    const float4* const tl[1] = {ps_tile(a.img, a.im.wq, wave - 2, nck, 5)};
    ps_gemv<1>(qo, sm + L.qv, tl, nck, lane);
    const int col = (wave - 2) * 16 + lc;
    if (lq == 0 && wave < 7) sm[L.qzv + col] = qo[0] + W[a.at_b[1] + col];
  }
  __syncthreads();
  PS_MARK(ps_ts_fwd, 3);

  // ---- phase 4: rows of the folded first attention layer's input, [k, q*k], k = [user state | item state | atten_info]
  {
    const int Dk4 = Dk >> 2;
    for (int e = tid; e < A * Dk4; e += PS_NT) {
      const int t = e / Dk4, j = (e - t * Dk4) * 4;
      const float* src = j < H ? sm + L.gout + t * H + j
                               : (j < 2 * H ? sm + L.gout + (MP + t) * H + (j - H) : sm + L.infos + t * 4 * K + (j - 2 * H));
      const float4 kv = *reinterpret_cast<const float4*>(src);
      const float4 qq = *reinterpret_cast<const float4*>(sm + L.qv + j);
      const float4 qk = make_float4(qq.x * kv.x, qq.y * kv.y, qq.z * kv.z, qq.w * kv.w);
      *reinterpret_cast<float4*>(sm + L.ainp + t * L.lda + j) = kv;
      *reinterpret_cast<float4*>(sm + L.ainp + t * L.lda + Dk + j) = qk;
      float* o = a.ainp + ((int64_t)b * A + t) * 2 * Dk;
      st4(o + j, kv);
      st4(o + Dk + j, qk);
    }
    const int padk = ps_up(2 * Dk, 16) - 2 * Dk;
    for (int e = tid; e < A * padk; e += PS_NT) {
      const int t = e / padk;
      sm[L.ainp + t * L.lda + 2 * Dk + (e - t * padk)] = 0.f;
    }
  }
  __syncthreads();
  PS_MARK(ps_ts_fwd, 4);

  // ---- phase 5: dense_3 (folded): a1 = relu([k, q*k] . Weff + qz)
  {
    const int nca = (2 * Dk + 15) >> 4;
    ps_f32x4 acc[1][MT];
    ps_zero<MT, 1>(acc);
    const float4* const tl[1] = {ps_tile(a.img, a.im.weff, wave, nca, 5)};
    ps_mma<MT, 1>(acc, sm + L.ainp, L.lda, tl, nca, lane);
    if (wave < 5) {
      const int col = wave * 16 + lc;
      const float qz = sm[L.qzv + col];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = m * 16 + 4 * lq + v;
          if (row < A) {
            const float x = fmaxf(acc[0][m][v] + qz, 0.f);
            sm[L.a1s + row * L.ld1 + col] = x;
            a.a1[((int64_t)b * A + row) * 80 + col] = x;
          }
        }
    }
  }
  __syncthreads();
  PS_MARK(ps_ts_fwd, 5);

  // ---- phase 6: dense_4: a2 = relu(a1 . W4 + b4)
  {
    ps_f32x4 acc[1][MT];
    ps_zero<MT, 1>(acc);
    const float4* const tl[1] = {ps_tile(a.img, a.im.w4, wave, 5, 3)};
    ps_mma<MT, 1>(acc, sm + L.a1s, L.ld1, tl, 5, lane);
    const int col = wave * 16 + lc;
    const float b4 = W[a.at_b[2] + min(col, 39)];
    if (wave < 3) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = m * 16 + 4 * lq + v;
          if (row < A && col < 40) {
            const float x = fmaxf(acc[0][m][v] + b4, 0.f);
            sm[L.a2s + row * L.ld2 + col] = x;
            a.a2[((int64_t)b * A + row) * 40 + col] = x;
          }
        }
    }
  }
  __syncthreads();
  PS_MARK(ps_ts_fwd, 6);

  // ---- phase 7: dense_5, where(mask, ., -2^32+1), softmax over the slices (score.py:177-185); a lane per slice
  if (wave == 0) {
    const bool tok = lane < A;
    const int t = tok ? lane : 0;
    float acc = 0.f;
    for (int n = 0; n < 40; ++n) acc = fmaf(sm[L.a2s + t * L.ld2 + n], W[a.at_w5 + n], acc);
    const float sv = (tok && lane < len) ? acc + W[a.at_b[3]] : -4294967295.0f;
    const float mx = wave_max(tok ? sv : -INFINITY);
    const float e = tok ? expf(sv - mx) : 0.f;
    const float den = wave_sum(e);
    const float p = e / den;
    if (tok) {
      sm[L.sc + t] = p;
      a.att_score[(int64_t)b * A + t] = p;
    }
  }
  __syncthreads();
  PS_MARK(ps_ts_fwd, 7);
  // pooled states sum_t rep_t * score_t (score.py:214-215) into the head's input
  if (tid < 2 * H) {
    const int side = tid / H, j = tid - side * H;
    float acc = 0.f;
    for (int t = 0; t < A; ++t) acc = fmaf(sm[L.gout + (side * MP + t) * H + j], sm[L.sc + t], acc);
    const int off = side ? s.off_i : s.off_u;
    if (off >= 0) {
      sm[L.hin + off + j] = acc;
      a.head_inp[(int64_t)b * Dh + off + j] = acc;
    }
  }
  __syncthreads();
  PS_MARK(ps_ts_fwd, 8);
  // bn1: inference-mode affine (score.py:69)
  for (int j = tid; j < Dh; j += PS_NT) {
    const float v = sm[L.hin + j] * (W[a.bn_g + j] * a.rs) + W[a.bn_b + j];
    sm[L.bns + j] = v;
    a.bn[(int64_t)b * Dh + j] = v;
  }
  __syncthreads();
  PS_MARK(ps_ts_fwd, 9);

  // ---- phase 8: fc1 200 relu dropout (13 column tiles: waves 0-4 take two)
  {
    const int nc = (Dh + 15) >> 4;
    float fo[2];
    const float4* const tl[2] = {ps_tile(a.img, a.im.fc1, wave, nc, 13), ps_tile(a.img, a.im.fc1, wave + 8, nc, 13)};
    ps_gemv<2>(fo, sm + L.bns, tl, nc, lane);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int col = (wave + 8 * i) * 16 + lc;
      if (lq == 0 && col < 200) {
        const float v = ps_act(fo[i], W[a.fc_b[0] + col], a.drop, a.keep, a.mask0, seed0, b, col, 200);
        sm[L.f1s + col] = v;
        a.f1[(int64_t)b * 200 + col] = v;
      }
    }
  }
  __syncthreads();
  PS_MARK(ps_ts_fwd, 10);
  // ---- phase 9: fc2 80 relu dropout
  {
    float fo[1];
    const float4* const tl[1] = {ps_tile(a.img, a.im.fc2, wave, 13, 5)};
    ps_gemv<1>(fo, sm + L.f1s, tl, 13, lane);
    const int col = wave * 16 + lc;
    if (lq == 0 && wave < 5) {
      const float v = ps_act(fo[0], W[a.fc_b[1] + col], a.drop, a.keep, a.mask1, seed1, b, col, 80);
      sm[L.f2s + col] = v;
      a.f2[(int64_t)b * 80 + col] = v;
    }
  }
  __syncthreads();
  PS_MARK(ps_ts_fwd, 11);
  // ---- phase 10: fc3, sigmoid, the sample's log-loss term and its gradient, dz2 (score.py:74-81)
  if (wave == 0) {
    float part = sm[L.f2s + lane] * W[a.fc_w3 + lane];
    if (lane < 16) part = fmaf(sm[L.f2s + 64 + lane], W[a.fc_w3 + 64 + lane], part);
    const float z = wave_sum(part) + W[a.fc_b[2]];
    const float p = sigmoidf_(z);
    const float lab = (float)a.label[b];
    const float eps = 1e-7f;
    const float dp = (-lab / (p + eps) + (1.0f - lab) / (1.0f - p + eps)) / (float)s.Bglobal;
    const float dl = dp * p * (1.0f - p);
    if (lane == 0) {
      a.logit[b] = z;
      a.y[b] = p;
      a.dlogit[b] = dl;
      // the sample's term goes through to memory (agent scope) and is acknowledged BEFORE this workgroup counts itself done:
      // whoever counts last then reads every term past its caches -- no cache-wide fence (a workgroup-count of L2 write-backs
      // at the end of every step slows whatever runs beside this kernel)
      __hip_atomic_store(&a.lossb[b], -lab * logf(p + eps) - (1.0f - lab) * logf(1.0f - p + eps), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int n = lane; n < 80; n += 64) {
      const float q = dl * W[a.fc_w3 + n] / a.keep;
      a.dz2[(int64_t)b * 80 + n] = sm[L.f2s + n] > 0.f ? q : 0.f;
    }
    if (lane == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned int t = __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      reinterpret_cast<volatile int*>(sm + L.misc)[0] = (t == (unsigned int)s.B - 1u) ? 1 : 0;
    }
  }
  ps_touch_use(warm, sm + L.misc + 4);
  PS_MARK(ps_ts_fwd, 12);
  __syncthreads();
  if (reinterpret_cast<volatile int*>(sm + L.misc)[0] == 0) return;
  // ---- the last workgroup: loss[1] = sum_b lossb / Bglobal, loss[2] = 0.5 * sum(parts), loss[0] = loss[1] + lambda * loss[2] -- the
  // sums of loss_final_kernel (head.hip), thread for thread, so the layer-by-layer pass gives the same bits
  {
    float* sh = sm;                  // (every phase is behind the barrier above: any region will do)
    float* sp = sh + 256;
    if (tid < 256) {
      float acc = 0.f;
      for (int i = tid; i < s.B; i += 256) acc += __hip_atomic_load(&a.lossb[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sh[tid] = acc;
      sp[tid] = a.part[tid];
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) { sh[tid] += sh[tid + o]; sp[tid] += sp[tid + o]; }
      __syncthreads();
    }
    if (tid == 0) {
      float l1 = sh[0] * a.inv_bglobal;
      const float l2 = 0.5f * sp[0];
      float l0 = l1 + a.lambda * l2;
      const int32_t bad = a.id_status ? __hip_atomic_load(a.id_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      if (bad) l0 = l1 = __int_as_float(0x7fc00000);
      a.loss[0] = l0; a.loss[1] = l1; a.loss[2] = l2; a.loss[3] = (float)bad;
      if (a.loss_host) {
        __hip_atomic_store(&a.loss_host[1], l1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&a.loss_host[2], l2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&a.loss_host[3], (float)bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&a.loss_host[0], l0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host side
int ps_plan_shape(int B, int A, int Tidx, int K, int D, int Fu, int Fi, int H, int NI, int Dk, int Dhead, int off_u, int off_i,
                  int off_ti, int off_tu, int Bglobal, PsShape* o) {
  if (B <= 0 || B > PS_MAX_B || A <= 0 || A > 48 || K <= 0 || K > 10 || H != 32 || (D & 3) || D <= 0) return SCORE_E_SHAPE;
  if ((Fu + Fi) * D > 192) return SCORE_E_SHAPE;         // (tile-to-wave deals of the kernels: <= 12 column tiles of I, <= 16 of Dhead)
  if (NI != 4 * K || Dk != 2 * H + NI) return SCORE_E_SHAPE;           // (SCORE / SCORE_USER / SCORE_ITEM)
  memset(o, 0, sizeof(*o));
  o->B = B; o->A = A; o->Tidx = Tidx; o->K = K; o->D4 = D / 4; o->Fu = Fu; o->Fi = Fi; o->H = H;
  o->Du = Fu * D; o->Di = Fi * D; o->I = o->Du + o->Di; o->NI = NI; o->Dk = Dk; o->Dhead = Dhead;
  o->off_u = off_u; o->off_i = off_i; o->off_ti = off_ti; o->off_tu = off_tu; o->MP = (A + 15) / 16 * 16; o->Bglobal = Bglobal;
  const int F[2] = {Fi, Fu};
  for (int c = 0; c < 2; ++c) {
    o->nslots[c] = F[c] * (D / 4);
    if (o->nslots[c] > 64) return SCORE_E_SHAPE;
    int gs = 1;
    while (gs < o->nslots[c]) gs <<= 1;
    o->GS[c] = gs;
  }
  o->V0 = (A * o->GS[0] + 63) / 64 * 64;
  o->Vtot = o->V0 + A * o->GS[1];
  PsLds L;
  ps_lds_layout(*o, &L);
  if ((size_t)L.fwd_total * 4 > 150 * 1024 || (size_t)L.bwd_total * 4 > 150 * 1024) return SCORE_E_SHAPE;
  return 0;
}

void ps_plan_images(const PsShape& s, PsImages* im) {
  int64_t cur = 0;
  auto take = [&](int K, int N) { int64_t o = cur; cur += ps_image_floats(K, N); return o; };
  const int H3 = 3 * s.H;
  for (int sd = 0; sd < 2; ++sd) im->wx[sd] = take(s.I, H3);
  im->q2 = take(s.I, s.Dk); im->wq = take(s.Dk, 80); im->weff = take(2 * s.Dk, 80); im->w4 = take(80, 40);
  im->fc1 = take(s.Dhead, 200); im->fc2 = take(200, 80);
  im->fc2t = take(80, 200); im->fc1t = take(200, s.Dhead); im->w4t = take(40, 80); im->wefft = take(80, 2 * s.Dk);
  im->wqt = take(80, s.Dk); im->q2t = take(s.Dk, s.I);
  for (int sd = 0; sd < 2; ++sd) im->wxt[sd] = take(H3, s.I);
  im->total = cur;
}

int score_launch_ps_prep(const PsPrepArgs& a_, hipStream_t s) {
  PsPrepArgs a = a_;
  int blocks = 0;
  for (int q = 0; q < a.njobs; ++q) {
    a.job[q].first_block = blocks;
    const int items = ((a.job[q].K + 15) / 16) * ((a.job[q].N + 15) / 16) * 64;
    blocks += (items + 255) / 256;
  }
  a.img_blocks = blocks;
  a.zero_blocks = a.zero ? (int)cdiv64(cdiv64(a.zero_floats, 4), 256) : 0;
  hipLaunchKernelGGL(ps_prep_kernel, dim3(blocks + PS_L2_PARTS + a.zero_blocks), dim3(256), 0, s, a);
  SCORE_CHECK_LAUNCH();
  return 0;
}

template <int KMAX, int MT>
static int ps_fwd_launch(const PsFwdArgs& a, size_t lds, hipStream_t s) {
  if (lds > 48 * 1024) {
    static bool done[SCORE_PS_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SCORE_PS_MAX_DEVICES) return SCORE_E_BADARG;
    if (!done[dev]) {
      hipError_t e = hipFuncSetAttribute((const void*)ps_fwd_kernel<KMAX, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
      done[dev] = true;
    }
  }
  hipLaunchKernelGGL((ps_fwd_kernel<KMAX, MT>), dim3(a.s.B), dim3(PS_NT), lds, s, a);
  SCORE_CHECK_LAUNCH();
  return 0;
}

int score_launch_ps_fwd(const PsFwdArgs& a, hipStream_t s) {
  PsLds L;
  ps_lds_layout(a.s, &L);
  const size_t lds = (size_t)L.fwd_total * 4;
  const int mt = a.s.MP / 16;
  if (a.s.K <= 5) {
    if (mt == 1) return ps_fwd_launch<5, 1>(a, lds, s);
    if (mt == 2) return ps_fwd_launch<5, 2>(a, lds, s);
    return ps_fwd_launch<5, 3>(a, lds, s);
  }
  if (mt == 1) return ps_fwd_launch<10, 1>(a, lds, s);
  if (mt == 2) return ps_fwd_launch<10, 2>(a, lds, s);
  return ps_fwd_launch<10, 3>(a, lds, s);
}

#if defined(PS_PHASE_TIMING)
extern "C" int score_ps_phase_read_fwd(unsigned long long* out32) {
  return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(ps_ts_fwd), 32 * sizeof(unsigned long long));
}
#endif
