// H = 128 recurrences on the bf16 matrix cores, fp32-accurate ("bf16x3", as gemm_bf16x3.hip).
//
// gru.hip's register-resident kernels feed v_mfma_f32_16x16x4_f32, which runs at the VALU rate: at H = 128 a
// step costs a SIMD 6,144 matrix-pipe cycles and the pipe is what the forward waits for (tools/gru_reg_probe.py:
// 46 of 88 us).  Here every fp32 operand is split exactly into three bf16 planes (8+8+8 significand bits) and a
// product is the six v_mfma_f32_16x16x32_bf16 whose weight is above 2^-24: 2,304 pipe cycles per step and SIMD,
// to one fp32 rounding of the exact product.  Same ownership as gru.hip (workgroup = 8 waves x 16 batch rows for
// all T steps, wave w owns columns 16w..16w+15 of r, u, the candidate and the state in every phase), so:
//  * the recurrent weights stay in registers as split planes (144 VGPRs per lane),
//  * the state h (forward) / the running dL/dh (backward) live in four registers per lane, fp32, never in LDS,
//  * LDS only holds the MFMA A operands, split once by the lane that produced the value: three [16][K] bf16
//    planes, k contiguous, row stride K+8 (16-B aligned, the 16 rows of a ds_read_b128 land on 64 banks).
// Semantics: score.py:205-208 (dynamic_rnn over GRUCell, sequence_length = length: state frozen and output 0 past
// the length), as gru.hip; results differ from gru.hip's in the last fp32 bits only (accumulation order).
#include <algorithm>
#include "common.h"
#include "kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int XH = 128, XNW = 8, XRB = 16;
#ifndef XG_PREFETCH
#define XG_PREFETCH 3        // steps of look-ahead of the forward kernel's L2 prefetch (0: none); 3: +0.7 % at cfg-3, five pairs
#endif
constexpr int XKS = XH / 32;          // k-steps of 32
constexpr int XLD = XH + 8;           // bf16 per LDS row of a [16][H] plane
constexpr int XLD2 = 2 * XH + 8;      //                      a [16][2H] plane

// tools/gru_x3_probe.py builds this file with one ingredient stripped at a time (wrong results, timing only)
#if defined(XGP_NOSTORE)
#define XG_STORE(lhs, v) do { if ((v) == 123.456f) lhs = (v); } while (0)
#else
#define XG_STORE(lhs, v) lhs = (v)
#endif
#if defined(XGP_NOXLOAD)
#define XG_LOAD(e) 0.5f
#else
#define XG_LOAD(e) (e)
#endif
#if defined(XGP_DIVRCP)
#define XG_RCP(x) __frcp_rn(x)              // correctly rounded: a ten-instruction division sequence
#else
#define XG_RCP(x) __builtin_amdgcn_rcpf(x)  // v_rcp_f32 (1 ulp)
#endif
__device__ __forceinline__ float x_sigmoid(float x) { return XG_RCP(1.0f + __expf(-x)); }
__device__ __forceinline__ float x_tanh(float x) { return 1.0f - 2.0f * XG_RCP(__expf(2.0f * x) + 1.0f); }

// exact 3-way split (truncation): the three bf16 bit patterns sit in the upper halves of h, m, l
__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  const uint32_t xb = __float_as_uint(x);
  h = xb & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xFFFF0000u;
  l = __float_as_uint(r1 - __uint_as_float(m));
}
__device__ __forceinline__ uint32_t pack2(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// eight consecutive-k fp32 values -> the three planes of one 16x16x32 operand fragment
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 (&out)[3]) {
  uint32_t p[3][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split3(x[j], p[0][j], p[1][j], p[2][j]);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    uint4 v;
    v.x = pack2(p[q][0], p[q][1]); v.y = pack2(p[q][2], p[q][3]);
    v.z = pack2(p[q][4], p[q][5]); v.w = pack2(p[q][6], p[q][7]);
    out[q] = __builtin_bit_cast(bf16x8, v);
  }
}
// one value into the three LDS planes (plane stride `ps` elements)
__device__ __forceinline__ void put3(unsigned short* base, int ps, int idx, float x) {
  uint32_t h, m, l;
  split3(x, h, m, l);
  base[idx] = (unsigned short)(h >> 16);
  base[ps + idx] = (unsigned short)(m >> 16);
  base[2 * ps + idx] = (unsigned short)(l >> 16);
}
__device__ __forceinline__ void get3(const unsigned short* base, int ps, int idx, bf16x8 (&out)[3]) {
#pragma unroll
  for (int q = 0; q < 3; ++q) out[q] = *reinterpret_cast<const bf16x8*>(base + q * ps + idx);
}
#if defined(XGP_NOMFMA)
#define X_MFMA(a, b, c) (c)
#else
#define X_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif
// c += A . B: the small terms first
__device__ __forceinline__ f32x4 mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 c) {
  c = X_MFMA(a[0], b[2], c);
  c = X_MFMA(a[2], b[0], c);
  c = X_MFMA(a[1], b[1], c);
  c = X_MFMA(a[0], b[1], c);
  c = X_MFMA(a[1], b[0], c);
  c = X_MFMA(a[0], b[0], c);
  return c;
}

// Addressing: base pointer + uniform byte offset (the time step: scalar registers) + a 32-bit per-lane byte offset
// fixed for the whole launch (row and column), i.e. the saddr + voffset form of global_load / global_store.  With
// 64-bit per-lane pointers the compiler kept one induction variable per array and row alive across the time loop
// (~70 VGPRs), spilt weight fragments to make room and reloaded them behind s_waitcnt vmcnt(0) every phase.
// (readfirstlane pins the uniform part in scalar registers and keeps loop strength reduction from folding the
// time step back into per-lane 64-bit induction variables)
__device__ __forceinline__ uint64_t uni_addr(const void* base, int64_t uni) {
  const uint64_t p = reinterpret_cast<uint64_t>(base) + (uint64_t)uni;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
  return ((uint64_t)hi << 32) | lo;
}
typedef __attribute__((address_space(1))) float gfloat;     // (an integer cast back to a plain pointer would be a flat access)
__device__ __forceinline__ float ldg(const float* base, int64_t uni, uint32_t voff) {
  return XG_LOAD(*(const gfloat*)(uni_addr(base, uni) + voff));
}
__device__ __forceinline__ gfloat* stp(float* base, int64_t uni, uint32_t voff) {
  return (gfloat*)(uni_addr(base, uni) + voff);
}

// (no predicated memory operation in the time loop: rows past the batch duplicate the last sample, see below)
__global__ __launch_bounds__(64 * XNW) void gru_fwd_x3_kernel(const GruArgs a) {
  constexpr int H = XH, PS = XRB * XLD;
  __shared__ __attribute__((aligned(16))) unsigned short hp[3 * PS], rp[3 * PS];
  const int tiles_b = (a.B + XRB - 1) / XRB;
  const int side = blockIdx.x / tiles_b;
  const GruSide& sd = a.s[side];
  const int b0 = (blockIdx.x - side * tiles_b) * XRB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, lq = lane >> 4;
  const int T = a.T;
  const int j = wave * 16 + lc;       // this lane's column of r, u, c and h

  bf16x8 wr[XKS][3], wu[XKS][3], wc[XKS][3];
#pragma unroll
  for (int s = 0; s < XKS; ++s) {
    float xr[8], xu[8], xc[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int64_t k = 32 * s + 8 * lq + jj;
      xr[jj] = sd.Wg[k * sd.ldwg + j];
      xu[jj] = sd.Wg[k * sd.ldwg + H + j];
      xc[jj] = sd.Wc[k * sd.ldwc + j];
    }
    split8(xr, wr[s]); split8(xu, wu[s]); split8(xc, wc[s]);
  }
  int len[4];
  bool rok[4];
  uint32_t rb3[4], rbo[4];        // byte offsets of (sample, t = 0, column j) in the [.,3H] arrays / in out
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + lq * 4 + r;
    rok[r] = true;                      // rows past the batch duplicate the last sample (same values to the same
    const int bc = min(b, a.B - 1);     // addresses: a benign race), so a ragged batch has no predicated access either   // samples past the batch read the last one's (never stored)
    len[r] = a.length[bc];
    rb3[r] = (uint32_t)(((int64_t)bc * T * 3 * H + j) * 4);
    rbo[r] = (uint32_t)(((int64_t)bc * T * sd.ldo + j) * 4);
  }
  for (int e = tid; e < 3 * PS; e += 64 * XNW) hp[e] = 0;
  float h[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  // x-projection of a step, read unconditionally (clamped) about one step ahead: each of its three parts is
  // fetched again right after its last use (r, u, candidate epilogue), into the registers it is consumed from
  float x[3][4];
  auto fetch_x = [&](int g, int t) {
    const int64_t u3 = ((int64_t)min(t, T - 1) * 3 * H + g * H) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) x[g][r] = ldg(sd.xproj, u3, rb3[r]);
  };
  fetch_x(0, 0); fetch_x(1, 0); fetch_x(2, 0);
#if XG_PREFETCH
  // The x-projection rows of step t + XG_PREFETCH pulled towards the L2 a few steps ahead of the loads above: ONE load per wave and
  // step whose lanes each touch one 128-B line (16 samples x 12 lines per step and workgroup = 24 lines per wave), its value
  // folded into a word nobody reads.  The loads above run one step ahead -- enough alone on the chip with warm caches (50 us), not
  // from HBM (66 us with the caches flushed, 54 with this) and not beside the occurrence sort on the other stream
  // (tools/gru_contention_probe.py).  The backward kernel's twin measured slower (profiles/r06_probes.md section 11).
  const int pfl = min(lane, 23), pfL = wave * 24 + pfl;
  const uint32_t pfoff = (uint32_t)(((int64_t)min(b0 + pfL / 12, a.B - 1) * T * 3 * H) * 4 + (pfL % 12) * 128);
  float pfv = 0.f;
  uint32_t pfsink = 0;
#endif
  const int aoff = lc * XLD + 8 * lq;
  // The first step is peeled: at the loop header the compiler's s_waitcnt placement merges the state of the
  // prologue (loads with nothing behind them) with the back edge's (loads with the step's sixteen stores behind
  // them) and keeps the stricter count, so every step waited for the previous step's stores to be acknowledged.
  auto step = [&](const int t) {
    const int64_t u3 = (int64_t)t * 3 * H * 4;
#if XG_PREFETCH
    pfsink ^= __float_as_uint(pfv);
    pfv = ldg(sd.xproj, (int64_t)min(t + XG_PREFETCH, T - 1) * 3 * H * 4, pfoff);
#endif
    // gates = sigmoid(xproj[:, :2H] + h . Wg)
    f32x4 ar = {0.f, 0.f, 0.f, 0.f}, au = {0.f, 0.f, 0.f, 0.f};
    {
      bf16x8 af[XKS][3];             // all of the phase's operand reads go out before its MFMA chain
#pragma unroll
      for (int s = 0; s < XKS; ++s) get3(hp, PS, aoff + 32 * s, af[s]);
#pragma unroll
      for (int s = 0; s < XKS; ++s) {
        ar = mfma6(af[s], wr[s], ar);
        au = mfma6(af[s], wu[s], au);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = lq * 4 + r;
      const float rg = x_sigmoid(ar[r] + x[0][r]);
      if (rok[r]) XG_STORE(*stp(sd.gates, u3, rb3[r]), rg);
      put3(rp, PS, i * XLD + j, rg * h[r]);
    }
    fetch_x(0, t + 1);
    __syncthreads();
    // c = tanh(xproj[:, 2H:] + (r*h) . Wc) ; h' = u*h + (1-u)*c.  The update gate's epilogue is not on the
    // r -> r*h -> barrier chain: it runs behind the barrier, in the shadow of the candidate product's operand reads.
    f32x4 ac = {0.f, 0.f, 0.f, 0.f};
    float u[4];
    {
      bf16x8 af[XKS][3];
#pragma unroll
      for (int s = 0; s < XKS; ++s) get3(rp, PS, aoff + 32 * s, af[s]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        u[r] = x_sigmoid(au[r] + x[1][r]);
        if (rok[r]) XG_STORE(*stp(sd.gates, u3 + H * 4, rb3[r]), u[r]);
      }
      fetch_x(1, t + 1);
#pragma unroll
      for (int s = 0; s < XKS; ++s) ac = mfma6(af[s], wc[s], ac);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = lq * 4 + r;
      const float c = x_tanh(ac[r] + x[2][r]);
      const float hn = u[r] * h[r] + (1.0f - u[r]) * c;
      const bool live = t < len[r];
      if (rok[r]) {
        XG_STORE(*stp(sd.gates, u3 + 2 * H * 4, rb3[r]), c);
        XG_STORE(*stp(sd.out, (int64_t)t * sd.ldo * 4, rbo[r]), (live ? hn : 0.f));
      }
      h[r] = live ? hn : h[r];
      put3(hp, PS, i * XLD + j, h[r]);
    }
    fetch_x(2, t + 1);
    __syncthreads();
  };
  step(0);
  for (int t = 1; t < T; ++t) step(t);
#if XG_PREFETCH
  asm volatile("" : : "v"(pfsink ^ __float_as_uint(pfv)));     // (keeps the prefetch loads alive: an empty statement that "reads" them)
#endif
  if (sd.final_state) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      sd.final_state[(int64_t)min(b0 + lq * 4 + r, a.B - 1) * H + j] = h[r];
  }
}

__global__ __launch_bounds__(64 * XNW) void gru_bwd_x3_kernel(const GruArgs a) {
  constexpr int H = XH, PS = XRB * XLD, PS2 = XRB * XLD2;
  __shared__ __attribute__((aligned(16))) unsigned short dpc[3 * PS], dpg[3 * PS2];
  // column sums of what each lane writes into dxproj (its columns j, H + j, 2H + j; rows of real samples only): the GRU
  // bias gradients' partial sums, so that nobody reads the [B*T, 3H] matrix again for them.  Kept in LDS, one private
  // slot per lane and gate (ds_add_f32, no contention): three more live registers spilt 39 in this kernel
  __shared__ float sbias[3][64 * XNW];
  const int tiles_b = (a.B + XRB - 1) / XRB;
  const int side = blockIdx.x / tiles_b;
  const GruSide& sd = a.s[side];
  const int b0 = (blockIdx.x - side * tiles_b) * XRB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  sbias[0][tid] = 0.f; sbias[1][tid] = 0.f; sbias[2][tid] = 0.f;
  const int lc = lane & 15, lq = lane >> 4;
  const int T = a.T;
  const int j = wave * 16 + lc;

  // B operands of the two transposed products: B[k][j] = Wc[j][k] (k < H), Wg[j][k] (k < 2H)
  bf16x8 wct[XKS][3], wgt[2 * XKS][3];
#pragma unroll
  for (int s = 0; s < 2 * XKS; ++s) {
    float xg[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) xg[jj] = sd.Wg[(int64_t)j * sd.ldwg + 32 * s + 8 * lq + jj];
    split8(xg, wgt[s]);
    if (s < XKS) {
      float xc[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) xc[jj] = sd.Wc[(int64_t)j * sd.ldwc + 32 * s + 8 * lq + jj];
      split8(xc, wct[s]);
    }
  }
  int len[4];
  bool rok[4];
  const int nreal = a.B - (b0 + lq * 4);      // rows r < nreal of this lane are real samples (the rest duplicate the last one)
  uint32_t rb3[4], rbh[4], rbo[4], rbd[4];   // byte offsets of (sample, t = 0, column j): [.,3H] / [.,H] arrays, out, dout
  float dh[4];                       // running dL/dh of this lane's four (row, column) elements
  const bool want_bias = sd.bias_slab != nullptr;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + lq * 4 + r;
    rok[r] = true;                      // rows past the batch duplicate the last sample (same values to the same
    const int bc = min(b, a.B - 1);     // addresses: a benign race), so a ragged batch has no predicated access either
    len[r] = a.length[bc];
    rb3[r] = (uint32_t)(((int64_t)bc * T * 3 * H + j) * 4);
    rbh[r] = (uint32_t)(((int64_t)bc * T * H + j) * 4);
    rbo[r] = (uint32_t)(((int64_t)bc * T * sd.ldo + j) * 4);
    rbd[r] = (uint32_t)(((int64_t)bc * T * sd.lddo + j) * 4);
    dh[r] = (sd.dfinal && rok[r]) ? sd.dfinal[(int64_t)bc * H + j] : 0.f;
  }
  // Saved activations of a step: read unconditionally (clamped addresses) about one step ahead.  Each array is
  // fetched again right after its last use in the step (u, c, dout: phase 1; r: phase 2), straight into the
  // registers it is consumed from -- only h_prev, which both phases and the stores need masked, keeps a copy.
  // (With a second copy of all five the kernel spilt three weight fragments, and every reload in the time loop
  // was an s_waitcnt vmcnt(0) on the step's own prefetch and stores.)
  float n_u[4], n_c[4], n_r[4], n_hp[4], n_do[4];
  auto fetch_ucd = [&](int t) {
    const int64_t tc = max(t, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      n_u[r] = ldg(sd.gates, (tc * 3 * H + H) * 4, rb3[r]); n_c[r] = ldg(sd.gates, (tc * 3 * H + 2 * H) * 4, rb3[r]);
      n_do[r] = ldg(sd.dout, tc * sd.lddo * 4, rbd[r]);
    }
  };
  auto fetch_r = [&](int t) {
    const int64_t tc = max(t, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) n_r[r] = ldg(sd.gates, tc * 3 * H * 4, rb3[r]);
  };
  auto fetch_hp = [&](int t) {
    const int64_t tp = max(t - 1, 0);      // (the value read for t = 0 is never used)
#pragma unroll
    for (int r = 0; r < 4; ++r) n_hp[r] = ldg(sd.out, tp * sd.ldo * 4, rbo[r]);
  };
  fetch_ucd(T - 1); fetch_r(T - 1); fetch_hp(T - 1);
  const int aoff = lc * XLD + 8 * lq, aoff2 = lc * XLD2 + 8 * lq;

  auto step = [&](const int t) {      // (first step peeled, as in the forward)
    float c_hp[4];                   // h_{t-1}, 0 past the length and at t = 0
#pragma unroll
    for (int r = 0; r < 4; ++r) c_hp[r] = (t < len[r] && t > 0) ? n_hp[r] : 0.f;
    __builtin_amdgcn_sched_barrier(0);     // (a fetch hoisted above the last use of its registers costs a copy
    fetch_hp(t - 1);                       //  behind a wait for the load just issued)
    // phase 1 (elementwise): dpu, dpc ; dh <- dh_tot * u      (results of a dead step are discarded by `live`)
    float su = 0.f, sc = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = lq * 4 + r;
      const bool live = t < len[r];
      const float u = n_u[r], c = n_c[r];
      const float d = dh[r] + n_do[r];
      const float du = d * (c_hp[r] - c), dc = d * (1.0f - u);
      const float v_dpu = live ? du * u * (1.0f - u) : 0.f;
      const float v_dpc = live ? dc * (1.0f - c * c) : 0.f;
      dh[r] = live ? d * u : dh[r];
      su += r < nreal ? v_dpu : 0.f;
      sc += r < nreal ? v_dpc : 0.f;
      if (rok[r]) {
        XG_STORE(*stp(sd.hprev, (int64_t)t * H * 4, rbh[r]), c_hp[r]);
        XG_STORE(*stp(sd.dxproj, ((int64_t)t * 3 * H + H) * 4, rb3[r]), v_dpu);
        XG_STORE(*stp(sd.dxproj, ((int64_t)t * 3 * H + 2 * H) * 4, rb3[r]), v_dpc);
      }
      put3(dpc, PS, i * XLD + j, v_dpc);
      put3(dpg, PS2, i * XLD2 + H + j, v_dpu);
    }
    if (want_bias) { atomicAdd(&sbias[1][tid], su); atomicAdd(&sbias[2][tid], sc); }
    __builtin_amdgcn_sched_barrier(0);
    fetch_ucd(t - 1);
    __syncthreads();
    // phase 2: d(rh) = dpc . Wc^T ; dpr = d(rh)*h_prev*r(1-r) ; dh += d(rh)*r
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < XKS; ++s) {
        bf16x8 af[3];
        get3(dpc, PS, aoff + 32 * s, af);
        acc = mfma6(af, wct[s], acc);
      }
      float sr = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = lq * 4 + r;
        const bool live = t < len[r];
        const float rr = live ? n_r[r] : 0.f, hp = c_hp[r];
        const float drh = acc[r];
        const float v_dpr = live ? drh * hp * rr * (1.0f - rr) : 0.f;
        dh[r] += live ? drh * rr : 0.f;
        sr += r < nreal ? v_dpr : 0.f;
        if (rok[r]) {
          XG_STORE(*stp(sd.dxproj, (int64_t)t * 3 * H * 4, rb3[r]), v_dpr);
          XG_STORE(*stp(sd.rh, (int64_t)t * H * 4, rbh[r]), (rr * hp));
        }
        put3(dpg, PS2, i * XLD2 + j, v_dpr);
      }
      if (want_bias) atomicAdd(&sbias[0][tid], sr);
      __builtin_amdgcn_sched_barrier(0);
      fetch_r(t - 1);
    }
    __syncthreads();
    // phase 3: dh += [dpr | dpu] . Wg^T
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2 * XKS; ++s) {
        bf16x8 af[3];
        get3(dpg, PS2, aoff2 + 32 * s, af);
        acc = mfma6(af, wgt[s], acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[r] += acc[r];
    }
    __syncthreads();
    };
  step(T - 1);
  for (int t = T - 2; t >= 0; --t) step(t);
  if (want_bias) {     // the four row groups of a column (lanes lc, lc + 16, lc + 32, lc + 48) in a fixed order
    float sb_r = sbias[0][tid], sb_u = sbias[1][tid], sb_c = sbias[2][tid];
    sb_r += __shfl_xor(sb_r, 16, 64); sb_u += __shfl_xor(sb_u, 16, 64); sb_c += __shfl_xor(sb_c, 16, 64);
    sb_r += __shfl_xor(sb_r, 32, 64); sb_u += __shfl_xor(sb_u, 32, 64); sb_c += __shfl_xor(sb_c, 32, 64);
    if (lq == 0) {
      float* o = sd.bias_slab + (int64_t)(blockIdx.x - side * tiles_b) * 3 * H;
      o[j] = sb_r; o[H + j] = sb_u; o[2 * H + j] = sb_c;
    }
  }
}

}  // namespace

bool score_gru_x3_ok(int H, int nw8) { return H == XH && nw8 != 0; }
// every per-lane byte offset must fit 32 bits
static bool x3_fits(const GruArgs& a, int nsides) {
  int64_t ld = 3 * XH;
  for (int i = 0; i < nsides; ++i) ld = std::max<int64_t>(ld, std::max(a.s[i].ldo, a.s[i].lddo));
  return (int64_t)a.B * a.T * ld * 4 < ((int64_t)1 << 32);
}

int score_gru_fwd_x3(GruArgs& a, int nsides, hipStream_t s) {
  if (!score_gru_x3_ok(a.H, a.nw8) || a.B <= 0 || a.T <= 0 || !x3_fits(a, nsides)) return SCORE_E_SHAPE;
  dim3 grid(nsides * ((a.B + XRB - 1) / XRB));
  hipLaunchKernelGGL(gru_fwd_x3_kernel, grid, dim3(64 * XNW), 0, s, a);
  SCORE_CHECK_LAUNCH();
  return 0;
}

int score_gru_bwd_x3(GruArgs& a, int nsides, hipStream_t s) {
  if (!score_gru_x3_ok(a.H, a.nw8) || a.B <= 0 || a.T <= 0 || !x3_fits(a, nsides)) return SCORE_E_SHAPE;
  dim3 grid(nsides * ((a.B + XRB - 1) / XRB));
  hipLaunchKernelGGL(gru_bwd_x3_kernel, grid, dim3(64 * XNW), 0, s, a);
  SCORE_CHECK_LAUNCH();
  bool slabs = true;
  for (int i = 0; i < nsides; ++i) slabs = slabs && a.s[i].bias_slab != nullptr;
  a.bias_slab_rows = slabs ? (a.B + XRB - 1) / XRB : 0;
  return 0;
}
