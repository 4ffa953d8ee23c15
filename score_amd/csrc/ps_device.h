// Device helpers shared by the per-sample whole-model kernels (ps_fwd.hip, ps_bwd.hip); see persample.h.
#pragma once
#include "persample.h"

typedef float ps_f32x4 __attribute__((ext_vector_type(4)));

// LDS carve-up (floats) of the two kernels, from the shape alone (host and device agree by construction)
struct PsLds {
  // forward
  int ldx, xs, xp, gout, infos, qs, qv, qzv, lda, ainp, ld1, a1s, ld2, a2s, sc, hin, bns, f1s, f2s, hs, misc, fwd_total;
  // backward (own layout; same row strides)
  int b_dz2, b_dz1, b_dh, b_x, b_sc, b_dsv, b_gout, b_a2, b_a1, b_dainp, b_kk, b_qv, b_dgru, b_dinfo, b_adz, b_dqd,
      b_dqv, b_dquery, lddx, b_dxp, b_dxs, b_hs, b_dzs, b_slab, b_f1, b_rs, bwd_total;
};
__host__ __device__ inline int ps_up(int x, int m) { return (x + m - 1) / m * m; }
__host__ __device__ inline void ps_lds_layout(const PsShape& s, PsLds* L) {
  const int H = s.H, A = s.A, MP = s.MP, I = s.I, Dk = s.Dk, Dh = s.Dhead, K = s.K;
  int cur = 0;
  auto take = [&](int n) { int o = cur; cur += ps_up(n, 4); return o; };
  L->ldx = ps_up(I, 16) + 4;
  L->lda = ps_up(2 * Dk, 16) + 4;
  L->ld1 = 80 + 4;
  L->ld2 = 48 + 4;
  // ---- forward.  [xs | xp] and [ainp | a1s | a2s] share one region: the first pair is dead once the recurrence has run
  const int szX = 2 * MP * L->ldx + 2 * A * 3 * H;
  const int szY = MP * L->lda + MP * L->ld1 + MP * L->ld2;
  const int u0 = take(szX > szY ? szX : szY);
  L->xs = u0; L->xp = u0 + 2 * MP * L->ldx;
  L->ainp = u0; L->a1s = u0 + MP * L->lda; L->a2s = L->a1s + MP * L->ld1;
  L->gout = take(2 * MP * H);
  L->infos = take(A * 4 * K);
  L->qs = take(ps_up(I, 16));
  L->qv = take(ps_up(Dk, 16));
  L->qzv = take(80);
  L->sc = take(MP);
  L->hin = take(ps_up(Dh, 16));
  L->bns = take(ps_up(Dh, 16));
  L->f1s = take(208);
  L->f2s = take(80);
  L->hs = take(4 * H);
  L->misc = take(16);
  L->fwd_total = cur;
  // ---- backward.  One region holds first [a2 | a1 | dainp | kk] (the attention's backward) and then, once those are dead,
  // [dxp (the recurrences' pre-activation gradients), later the co-attention's dW slabs | dxs]
  cur = 0;
  L->b_dz2 = take(80); L->b_dz1 = take(208); L->b_dh = take(ps_up(Dh, 16));
  L->b_x = take(ps_up(Dh, 16)); L->b_f1 = take(208);
  L->b_sc = take(MP); L->b_dsv = take(MP);
  L->b_gout = take(2 * A * H);
  L->lddx = 3 * H + 4;
  {
    const int szA = MP * L->ld2 + MP * L->ld1 + A * L->lda + A * Dk;
    const int slab = 2 * 8 * PS_NT;         // per call: (PS_NT / GS) groups x 2 Dx floats <= 8 * PS_NT
    const int dxp = 2 * MP * L->lddx;
    const int szB = (dxp > slab ? dxp : slab) + 2 * A * I;
    const int r0 = take(szA > szB ? szA : szB);
    L->b_a2 = r0; L->b_a1 = L->b_a2 + MP * L->ld2; L->b_dainp = L->b_a1 + MP * L->ld1; L->b_kk = L->b_dainp + A * L->lda;
    L->b_dxp = r0; L->b_slab = r0; L->b_dxs = r0 + (dxp > slab ? dxp : slab);
  }
  L->b_qv = take(ps_up(Dk, 16));
  L->b_dgru = take(2 * A * H);
  L->b_dinfo = take(A * 4 * K);
  L->b_rs = take(2 * A * K);
  L->b_adz = take(80); L->b_dqd = take(ps_up(Dk, 16)); L->b_dqv = take(ps_up(Dk, 16)); L->b_dquery = take(ps_up(I, 16));
  L->b_hs = take(6 * H);
  L->b_dzs = take(2 * MP);
  L->bwd_total = cur;
}

// acc[m] (rows 16 m + 4 lq + v, column 16 ct + lc) += A[rows][k] . image(ct): A in LDS with row stride lda (0: every row is
// row 0 -- a product with ONE valid row), zero in its columns [K, 16 nchunk); img_ct = the tile's first float4.
// The B operands of the next eight chunks are requested before the current eight are consumed.
template <int MT>
__device__ __forceinline__ void ps_mma(ps_f32x4 (&acc)[MT], const float* A, int lda, const float4* __restrict__ img_ct, int nchunk,
                                       int lane) {
  const int lc = lane & 15, lq = lane >> 4;
  const float* arow = A + lc * lda + 4 * lq;
  constexpr int U = 8;
  float4 bc[U], bn[U];
#pragma unroll
  for (int u = 0; u < U; ++u) bc[u] = img_ct[(u < nchunk ? u : nchunk - 1) * 64 + lane];
  for (int c0 = 0; c0 < nchunk; c0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + U + u;
      bn[u] = img_ct[(c < nchunk ? c : nchunk - 1) * 64 + lane];     // clamped, unconditional
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (c0 + u < nchunk) {      // (wave-uniform)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float4 av = *reinterpret_cast<const float4*>(arow + m * 16 * lda + (c0 + u) * 16);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bc[u].x, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bc[u].y, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bc[u].z, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bc[u].w, acc[m], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) bc[u] = bn[u];
  }
}
template <int MT>
__device__ __forceinline__ void ps_zero(ps_f32x4 (&acc)[MT]) {
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = ps_f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ const float4* ps_tile(const float* img, int64_t off, int ct, int nchunk) {
  return reinterpret_cast<const float4*>(img + off) + (int64_t)ct * nchunk * 64;
}

// fast transcendental forms of the recurrence epilogues (v_exp_f32 / v_rcp_f32, as csrc/gru.hip's register kernels)
__device__ __forceinline__ float ps_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ps_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }
// LDS traffic between the lanes of ONE wave: the wave's LDS operations execute in program order, this only keeps the
// compiler from moving them across
__device__ __forceinline__ void ps_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
