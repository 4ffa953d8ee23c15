// Shared device/host helpers for libscore_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/score_hip.h"

#define SCORE_WAVE 64

// Timing probes (tools/*_probe.py) compile single kernels with one ingredient stripped -- no loads, no matrix
// instruction, no stores: WRONG results by design.  Such a switch only compiles with -DSCORE_PROBE_BUILD beside it, which
// score_amd/build.py refuses to pass: the product library cannot carry one by accident.
#if (defined(X3_PROBE_NOLOADA) || defined(X3_PROBE_NOLOADB) || defined(X3_PROBE_NOSPLIT) || defined(X3_PROBE_NOLDSW) ||      \
     defined(X3_PROBE_NOLDSR) || defined(X3_PROBE_NOMFMA) || defined(X3_PROBE_NOSTORE) || defined(PANEL_PROBE_NOMFMA) ||      \
     defined(PANEL_PROBE_NOALOAD) || defined(PANEL_PROBE_NOASTORE) || defined(PANEL_PROBE_NOBLOAD) ||                         \
     defined(PANEL_PROBE_NOREAD) || defined(PANEL_PROBE_NOCSTORE) || defined(GRP_NOMFMA) || defined(GRP_NOSTORE) ||           \
     defined(GRP_NOXLOAD) || defined(GSP_NOMFMA) || defined(GSP_NOBLOAD) || defined(GSP_NOSTORE) || defined(GSP_NOXLOAD) ||   \
     defined(XGP_NOMFMA) || defined(XGP_NOSTORE) || defined(XGP_NOXLOAD) || defined(AFP_NOPRELOAD) || defined(AFP_NOMFMA) ||  \
     defined(AFP_NOEPI) || defined(AFP_NOBUILD) || defined(AFP_NOTAIL) || defined(AFP_NOINP)) &&                              \
    !defined(SCORE_PROBE_BUILD)
#error "a wrong-by-design probe switch needs -DSCORE_PROBE_BUILD (tools/*_probe.py pass it; score_amd/build.py never does)"
#endif

#define SCORE_CHECK_LAUNCH()                      \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

#define SCORE_TRY(expr)                           \
  do {                                            \
    int rc__ = (expr);                            \
    if (rc__ != 0) return rc__;                   \
  } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t align_up64(int64_t a, int64_t b) { return cdiv64(a, b) * b; }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// loads through a pointer the compiler cannot prove global (e.g. one read back from LDS): say so, or it emits
// flat_load, which also counts against lgkmcnt and so serialises with every LDS access around it
typedef float score_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_global(const float* p) {
  const score_v4f v = *(const __attribute__((address_space(1))) score_v4f*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float ld1_global(const float* p) { return *(const __attribute__((address_space(1))) float*)p; }
__device__ __forceinline__ float dot4(float4 a, float4 b) {
  return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 acc) {
  acc.x = fmaf(s, a.x, acc.x); acc.y = fmaf(s, a.y, acc.y);
  acc.z = fmaf(s, a.z, acc.z); acc.w = fmaf(s, a.w, acc.w);
  return acc;
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
// tf.train.AdamOptimizer's ApplyAdam on one element (TF training_ops): m += (g - m)(1-b1); v += (g*g - v)(1-b2);
// var -= m*alpha / (sqrt(v) + eps).  The contractions are spelled out so that every kernel applying it (the dense
// sweeps in head.hip, the time-tiled ones in adam_tiled.hip) rounds the same way whatever the compiler would choose.
__device__ __forceinline__ void score_adam1(float& p, float& m, float& v, float g, float omb1, float omb2, float alpha,
                                            float eps) {
  m = __builtin_fmaf(g - m, omb1, m);
  v = __builtin_fmaf(__builtin_fmaf(g, g, -v), omb2, v);
#if defined(SCORE_ADAM_IEEE_DIV)
  p = p - (m * alpha) / (sqrtf(v) + eps);
#else
  // v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the correctly rounded sqrtf and division (~25 instructions): the
  // zero-gradient replay of the time-tiled optimizer is one of these per owed step and element and was VALU-bound
  // (the window slice: 147 MB in 170 - 200 us).  The update term moves by <= 2 ulp of itself, far inside what the
  // order of the sums feeding g already varies; every Adam kernel goes through here, so they still agree bit for bit.
  p = p - (m * alpha) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + eps);
#endif
}
// ApplyAdam over a flat range of floats (the dense variables; score.py:96-99) by virtual block `blk` of `nblk`: the body of
// adam_kernel (head.hip) and of the dense half of adam_step_kernel (adam_tiled.hip), so both round alike.
__device__ __forceinline__ void score_adam_dense_body(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                      const float* __restrict__ g, int64_t n4, int64_t n, int64_t n_reg, float l2,
                                                      float alpha, float omb1, float omb2, float eps, int blk, int nblk) {
  int64_t i = (int64_t)blk * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)nblk * blockDim.x;
  for (; i < n4; i += stride) {
    float4 mm = ld4(m + i * 4), vv = ld4(v + i * 4), gg = ld4(g + i * 4);
    int64_t e = i * 4;
    // exact shortcut: with g = m = v = 0 (a row no batch has touched yet, no L2 term on it) ApplyAdam
    // leaves m, v and the variable bit-identical -- skip the variable's read and all three writes
    if (e >= n_reg && gg.x == 0.f && gg.y == 0.f && gg.z == 0.f && gg.w == 0.f && mm.x == 0.f && mm.y == 0.f &&
        mm.z == 0.f && mm.w == 0.f && vv.x == 0.f && vv.y == 0.f && vv.z == 0.f && vv.w == 0.f)
      continue;
    float4 pp = ld4(p + i * 4);
    if (e < n_reg) {  // d/dw of lambda * sum(w^2)/2   (build_l2norm, score.py:91-94)
      gg.x = e + 0 < n_reg ? fmaf(l2, pp.x, gg.x) : gg.x;
      gg.y = e + 1 < n_reg ? fmaf(l2, pp.y, gg.y) : gg.y;
      gg.z = e + 2 < n_reg ? fmaf(l2, pp.z, gg.z) : gg.z;
      gg.w = e + 3 < n_reg ? fmaf(l2, pp.w, gg.w) : gg.w;
    }
    score_adam1(pp.x, mm.x, vv.x, gg.x, omb1, omb2, alpha, eps);
    score_adam1(pp.y, mm.y, vv.y, gg.y, omb1, omb2, alpha, eps);
    score_adam1(pp.z, mm.z, vv.z, gg.z, omb1, omb2, alpha, eps);
    score_adam1(pp.w, mm.w, vv.w, gg.w, omb1, omb2, alpha, eps);
    st4(p + i * 4, pp); st4(m + i * 4, mm); st4(v + i * 4, vv);
  }
  // tail (n not a multiple of 4)
  if (blk == 0 && threadIdx.x < (unsigned)(n - n4 * 4)) {
    int64_t e = n4 * 4 + threadIdx.x;
    float pp = p[e], mm = m[e], vv = v[e], gg = g[e];
    if (e < n_reg) gg = fmaf(l2, pp, gg);
    score_adam1(pp, mm, vv, gg, omb1, omb2, alpha, eps);
    p[e] = pp; m[e] = mm; v[e] = vv;
  }
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// sum over the `gs` (power of two <= 64) consecutive lanes of a group; every lane gets the sum
__device__ __forceinline__ float group_sum(float v, int gs) {
  for (int off = gs >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, SCORE_WAVE);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum(v, SCORE_WAVE); }

// N independent group sums at once (same contract as group_sum for each v[n]; the result is bitwise the
// xor-butterfly's, so every lane of a group holds identical bits).  The four steps inside a row of 16 lanes
// are DPP adds (quad_perm xor 1, xor 2, then row_half_mirror / row_mirror, which reach the partner quad /
// half once the quads are uniform): one VALU instruction each, no LDS.  The two cross-row steps go through
// ds_bpermute with the N values of a step in flight together and the partner address computed once.
// A __shfl_xor chain per value costs ~9 VALU instructions and one dependent LDS round trip per step and value
// (66 round trips for the 11 scores of a K = 10 unit: that, not memory, bounded the fused gather).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int N>
__device__ __forceinline__ void group_sum_n(float (&v)[N], int gs) {
  if (gs >= 2) {
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] += dpp_f32<0xB1>(v[n]);     // quad_perm [1,0,3,2]
  }
  if (gs >= 4) {
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] += dpp_f32<0x4E>(v[n]);     // quad_perm [2,3,0,1]
  }
  if (gs >= 8) {
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] += dpp_f32<0x141>(v[n]);    // row_half_mirror: lane i <-> 7 - i
  }
  if (gs >= 16) {
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] += dpp_f32<0x140>(v[n]);    // row_mirror: lane i <-> 15 - i
  }
  if (gs >= 32) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int a16 = (lane ^ 16) << 2, a32 = (lane ^ 32) << 2;
    float t[N];
#pragma unroll
    for (int n = 0; n < N; ++n) t[n] = __int_as_float(__builtin_amdgcn_ds_bpermute(a16, __float_as_int(v[n])));
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] += t[n];
    if (gs >= 64) {
#pragma unroll
      for (int n = 0; n < N; ++n) t[n] = __int_as_float(__builtin_amdgcn_ds_bpermute(a32, __float_as_int(v[n])));
#pragma unroll
      for (int n = 0; n < N; ++n) v[n] += t[n];
    }
  }
}
__device__ __forceinline__ float wave_max(float v) {
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, SCORE_WAVE));
  return v;
}

// Logical tile (bx, by, bz) of this workgroup in a 1-D launch of gx*gy*gz workgroups.  Workgroups are dealt
// round-robin over the 8 XCDs (each with a private L2), so neighbours in launch order never share an L2;
// this hands every XCD one contiguous run of logical tiles, x fastest, so the tiles that stream the same
// operand panel (same by/bz) find it in one L2.  Bijective for any workgroup count; placement is a speed
// assumption only.
__device__ __forceinline__ void xcd_tile_coords_n(int nwg, int orig, int gx, int gy, int& bx, int& by, int& bz) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  bx = wg % gx;
  const int t = wg / gx;
  by = t % gy;
  bz = t / gy;
}
__device__ __forceinline__ void xcd_tile_coords(int gx, int gy, int& bx, int& by, int& bz) {
  xcd_tile_coords_n((int)gridDim.x, (int)blockIdx.x, gx, gy, bx, by, bz);
}

// counter-based uniform in [0,1): splitmix64 finaliser of (seed, idx); same value
// wherever and whenever it is evaluated, so backward never needs a stored mask.
__device__ __forceinline__ float hash_uniform(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

