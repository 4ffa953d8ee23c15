// Device helpers shared by the per-sample whole-model kernels (ps_fwd.hip, ps_bwd.hip); see persample.h.
#pragma once
#include "persample.h"

typedef float ps_f32x4 __attribute__((ext_vector_type(4)));

// tools/ps_phase_probe.py builds a variant of the library with -DPS_PHASE_TIMING: one workgroup in the middle of the grid
// then leaves a 100-MHz timestamp at every phase boundary (score_ps_phase_read returns them).  Nothing in the product build.
#if defined(PS_PHASE_TIMING)
#define PS_MARK(arr, i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) (arr)[i] = wall_clock64(); } while (0)
#else
#define PS_MARK(arr, i) do { } while (0)
#endif

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

// acc[i][m] (rows 16 m + 4 lq + v, column 16 ct_i + lc) += A[rows][k] . image(tile i) for NT column tiles of one product at once:
// A in LDS with row stride lda (0: every row is row 0 -- a product with ONE valid row), zero in its columns [K, 16 nchunk);
// tile[i] = first float4 of the tile's image, nullptr = no such tile (its loads go to a valid tile, its MFMAs are skipped).
// The A fragment of a chunk is read once for all tiles; the B operands of up to sixteen (chunk, tile) pairs are requested
// together before the first is consumed: a phase of the per-sample kernels is ONE memory round trip deep wherever its tiles
// can be dealt to the eight waves in one go (the latency of that round trip, not bytes or flops, is what a phase costs).
template <int MT, int NT>
__device__ __forceinline__ void ps_mma(ps_f32x4 (&acc)[NT][MT], const float* A, int lda, const float4* const (&tile)[NT], int nchunk,
                                       int lane) {
  constexpr int CB = 16 / NT;          // chunks per batch of loads
  const int lc = lane & 15, lq = lane >> 4;
  const float* arow = A + lc * lda + 4 * lq;
  const float4* tp[NT];
  bool tv[NT];
  const float4* any = tile[0];
#pragma unroll
  for (int i = 1; i < NT; ++i) any = any ? any : tile[i];
  if (!any) return;                    // (no tile for this wave)
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    tv[i] = tile[i] != nullptr;
    tp[i] = (tv[i] ? tile[i] : any) + lane;
  }
  for (int c0 = 0; c0 < nchunk; c0 += CB) {
    float4 bb[CB][NT];
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      const int c = c0 + u < nchunk ? c0 + u : nchunk - 1;       // clamped, unconditional
#pragma unroll
      for (int i = 0; i < NT; ++i) bb[u][i] = tp[i][c * 64];
    }
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      if (c0 + u < nchunk) {      // (wave-uniform)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float4 av = *reinterpret_cast<const float4*>(arow + m * 16 * lda + (c0 + u) * 16);
#pragma unroll
          for (int i = 0; i < NT; ++i) {
            if (tv[i]) {
              acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bb[u][i].x, acc[i][m], 0, 0, 0);
              acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bb[u][i].y, acc[i][m], 0, 0, 0);
              acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bb[u][i].z, acc[i][m], 0, 0, 0);
              acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bb[u][i].w, acc[i][m], 0, 0, 0);
            }
          }
        }
      }
    }
  }
}
// out[i] (every lane: column 16 ct_i + lc) = x[k] . image(tile i): the ONE-row products of the kernels (the query branch, the head,
// their backward) on the vector ALU.  v_mfma_f32_16x16x4_f32 computes sixteen rows whatever the number of valid ones, and the
// f32 matrix pipe is only twice the vector rate: one valid row of sixteen made fc1 (176 x 200) 2,800 pipe cycles per wave.  Same
// images, same k dealing as ps_mma: a lane multiplies its four k values of a chunk, the four lane quarters are added at the end.
template <int NT>
__device__ __forceinline__ void ps_gemv(float (&out)[NT], const float* x, const float4* const (&tile)[NT], int nchunk, int lane) {
  constexpr int CB = 16 / NT;
  const int lq = lane >> 4;
  const float* xq = x + 4 * lq;
  const float4* tp[NT];
  const float4* any = tile[0];
#pragma unroll
  for (int i = 1; i < NT; ++i) any = any ? any : tile[i];
#pragma unroll
  for (int i = 0; i < NT; ++i) out[i] = 0.f;
  if (!any) return;
#pragma unroll
  for (int i = 0; i < NT; ++i) tp[i] = (tile[i] ? tile[i] : any) + lane;
  float2 p[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) p[i] = make_float2(0.f, 0.f);
  for (int c0 = 0; c0 < nchunk; c0 += CB) {
    float4 bb[CB][NT];
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      const int c = c0 + u < nchunk ? c0 + u : nchunk - 1;
#pragma unroll
      for (int i = 0; i < NT; ++i) bb[u][i] = tp[i][c * 64];
    }
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      if (c0 + u < nchunk) {
        const float4 av = *reinterpret_cast<const float4*>(xq + (c0 + u) * 16);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          p[i].x = fmaf(av.x, bb[u][i].x, p[i].x); p[i].y = fmaf(av.y, bb[u][i].y, p[i].y);
          p[i].x = fmaf(av.z, bb[u][i].z, p[i].x); p[i].y = fmaf(av.w, bb[u][i].w, p[i].y);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    float v = p[i].x + p[i].y;
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    out[i] = tile[i] ? v : 0.f;
  }
}
// element i (0 / 1) of a two-entry array of the kernel arguments: a select, not a dynamic index (which sends the whole
// by-value argument struct through scratch memory)
#define PS2(arr, i) ((i) ? (arr)[1] : (arr)[0])
// v or zero, component by component (a ?: on the float4 STRUCT becomes a pointer select and a load through it: the
// gathered rows then live in scratch memory, one dependent round trip per neighbour)
__device__ __forceinline__ float4 ps_sel4(bool c, float4 v) {
  return make_float4(c ? v.x : 0.f, c ? v.y : 0.f, c ? v.z : 0.f, c ? v.w : 0.f);
}
template <int MT, int NT>
__device__ __forceinline__ void ps_zero(ps_f32x4 (&acc)[NT][MT]) {
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[i][m] = ps_f32x4{0.f, 0.f, 0.f, 0.f};
}
// first float4 of column tile ct of the image at float offset `off` (nullptr when the product has no such tile)
__device__ __forceinline__ const float4* ps_tile(const float* img, int64_t off, int ct, int nchunk, int nct = 1 << 30) {
  return ct < nct ? reinterpret_cast<const float4*>(img + off) + (int64_t)ct * nchunk * 64 : nullptr;
}

// One dword of every 128-byte line of img[0, floats) by this workgroup (NPF loads per thread, clamped): the weight images a
// kernel reads in its later phases are then in this XCD's L2 when they are asked for.  A phase of these kernels is one or
// two memory round trips deep, and a first touch through the fabric is ~2 us of it (measured: the same phase times with
// 8 and with 400 workgroups -- latency, not contention).  The caller keeps the returned value alive to the kernel's end
// (ps_touch_use), so nothing waits for these loads but the loads behind them.
#define PS_NPF 10
struct PsTouch { float w[PS_NPF]; };
__device__ __forceinline__ PsTouch ps_touch(const float* img, int64_t floats, int tid) {
  PsTouch t;
  const int64_t last = floats > 0 ? floats - 1 : 0;
#pragma unroll
  for (int u = 0; u < PS_NPF; ++u) {
    const int64_t o = ((int64_t)tid + (int64_t)u * PS_NT) * 32;
    t.w[u] = ld1_global(img + (o < last ? o : last));
  }
  return t;
}
__device__ __forceinline__ void ps_touch_use(const PsTouch& t, float* sink) {
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < PS_NPF; ++u) acc += t.w[u];
  if (acc == 1.2345678e-30f) *sink = acc;        // (never true for finite weights' sums in practice; keeps the loads)
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
