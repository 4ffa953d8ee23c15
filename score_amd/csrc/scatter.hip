// Embedding-gradient scatter without float atomics: "sort once, pull per row".
//
// Every use of a table row in a batch (an *occurrence*: a neighbour feature id in one
// of the four [B,T,K,F] tensors or a target feature id) contributes
//     a * G[bt, f*D:(f+1)*D] + b * w[f*D:(f+1)*D]
// to that row's gradient, where G is a [B*T, .] activation-gradient matrix that is tiny
// next to the row traffic (it stays in L2 / Infinity Cache) and (a, b) are two scalars
// per (unit, neighbour) written by coattn_bwd.  So instead of materialising R*B row
// gradients and adding them with HBM float atomics (1.3 TB/s chip-wide, 14x slower on the
// hot categorical rows), the occurrences are radix-sorted by row id once per batch and a
// group of D/4 lanes walks a window of the sorted list, accumulating in registers and
// storing each row gradient exactly once.  Runs that cross a window edge leave partial sums
// that a second kernel adds in window order, so the result is bitwise reproducible.
#include <cstring>
#include <cstdlib>
#include <rocprim/device/device_radix_sort.hpp>
#include "common.h"
#include "kernels.h"

// occurrence descriptor: seg[31:29] f[28:26] k[25:21] bt[20:0]
#define DESC(seg, f, k, bt) (((uint32_t)(seg) << 29) | ((uint32_t)(f) << 26) | ((uint32_t)(k) << 21) | (uint32_t)(bt))

__global__ void plan_fill_kernel(PlanFillArgs a, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.off[6]) return;
  int seg = 0;
#pragma unroll
  for (int s = 1; s < 6; ++s) seg += (i >= a.off[s]) ? 1 : 0;
  int64_t local = i - a.off[seg];
  int F = a.F[seg];
  uint32_t f, k, bt;
  if (seg < 4) {
    f = (uint32_t)(local % F);
    int64_t q = local / F;
    k = (uint32_t)(q % a.K);
    bt = (uint32_t)(q / a.K);
  } else {
    f = (uint32_t)(local % F);
    k = 0;
    bt = (uint32_t)(local / F);
  }
  uint32_t row = (uint32_t)a.idx[seg][local];
  uint32_t key = row;
  if (a.G > 1) key = ((row % a.G) << a.shift) | (row / a.G);   // (owner, local row)
  keys[i] = key;
  vals[i] = DESC(seg, f, k, bt);
}

int score_plan_temp_bytes(int64_t n, int end_bit, size_t* bytes) {
  uint32_t* nul = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, *bytes, nul, nul, nul, nul, (size_t)n, 0u, (unsigned)end_bit,
                                           (hipStream_t)0);
  return e == hipSuccess ? 0 : (int)e;
}

// keys_out/vals_out <- occurrences of the batch sorted by (owner, row)
int score_launch_plan(const PlanFillArgs& a, int key_bits, uint32_t* keys_in, uint32_t* vals_in, uint32_t* keys_out,
                      uint32_t* vals_out, void* temp, size_t temp_bytes, hipStream_t s) {
  int64_t n = a.off[6];
  hipLaunchKernelGGL(plan_fill_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, a, keys_in, vals_in);
  SCORE_CHECK_LAUNCH();
  size_t need = 0;
  SCORE_TRY(score_plan_temp_bytes(n, key_bits, &need));
  if (need > temp_bytes) return SCORE_E_WORKSPACE;
  hipError_t e = rocprim::radix_sort_pairs(temp, need, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                           (unsigned)key_bits, s);
  return e == hipSuccess ? 0 : (int)e;
}

// ------------------------------------------------------------------ pull
__device__ __forceinline__ float4 pull_contrib(const PullArgs& a, uint32_t desc, int ch4) {
  const int seg = desc >> 29, f = (desc >> 26) & 7, k = (desc >> 21) & 31;
  const int64_t bt = desc & 0x1FFFFF;
  const int col = f * a.D + ch4;
  float4 g = ld4(a.G[seg] + bt * a.ldg[seg] + a.gcol[seg] + col);
  float ca = a.cA[seg] ? a.cA[seg][bt * a.K + k] : a.constA[seg];
  float4 r = make_float4(ca * g.x, ca * g.y, ca * g.z, ca * g.w);
  if (a.cB[seg]) r = fma4(a.cB[seg][bt * a.K + k], ld4(a.Wv[seg] + col), r);
  return r;
}

__device__ __forceinline__ int64_t key_to_row(uint32_t key, int G, int shift) {
  if (G <= 1) return key;
  return (int64_t)(key & ((1u << shift) - 1)) * G + (key >> shift);
}

__global__ __launch_bounds__(256) void pull_kernel(const PullArgs a, const uint32_t* __restrict__ keys,
                                                   const uint32_t* __restrict__ vals, int64_t n, int WS,
                                                   float* __restrict__ out, float* __restrict__ pfirst,
                                                   float* __restrict__ plast) {
  const int LPR = a.LPRp;                                // lanes per group (power of two >= D/4)
  const int gpb = blockDim.x / LPR;                      // groups per block
  const int64_t w = (int64_t)blockIdx.x * gpb + threadIdx.x / LPR;
  const int ch4 = (threadIdx.x % LPR) * 4;
  const int64_t start = w * WS;
  if (start >= n || ch4 >= a.D) return;
  const int64_t end = start + WS < n ? start + WS : n;
  uint32_t cur = keys[start];
  const bool first_open = start > 0 && keys[start - 1] == cur;
  bool is_first = true;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = start; i < end; ++i) {
    uint32_t key = keys[i];
    if (key != cur) {
      if (cur != 0) {
        if (is_first && first_open) st4(pfirst + w * a.D + ch4, acc);
        else st4(out + key_to_row(cur, a.Gsh, a.shift) * a.D + ch4, acc);
      }
      acc = make_float4(0.f, 0.f, 0.f, 0.f);
      cur = key;
      is_first = false;
    }
    if (key != 0) acc = add4(acc, pull_contrib(a, vals[i], ch4));
  }
  if (cur != 0) {
    const bool open_right = end < n && keys[end] == cur;
    if (is_first && first_open) st4(pfirst + w * a.D + ch4, acc);
    else if (open_right) st4(plast + w * a.D + ch4, acc);
    else st4(out + key_to_row(cur, a.Gsh, a.shift) * a.D + ch4, acc);
  }
}

// first index in [lo, n) whose key differs from `key` (keys ascending, keys[lo-1] == key)
__device__ __forceinline__ int64_t run_end(const uint32_t* __restrict__ keys, int64_t lo, int64_t n, uint32_t key) {
  int64_t hi = n;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (keys[mid] == key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// A run that starts in window w and continues to the right: total = plast[w] + pfirst[w+1] + ...
// (window order, so the sum is reproducible).  Chains longer than LONG_CHAIN windows (the hot
// categorical rows: thousands of windows) are queued for pull_long_kernel.
#define LONG_CHAIN 16
__global__ __launch_bounds__(256) void pull_fixup_kernel(const PullArgs a, const uint32_t* __restrict__ keys,
                                                         int64_t n, int WS, float* __restrict__ out,
                                                         const float* __restrict__ pfirst,
                                                         const float* __restrict__ plast,
                                                         int* __restrict__ long_count, int2* __restrict__ long_list) {
  const int LPR = a.LPRp;
  const int gpb = blockDim.x / LPR;
  const int64_t w = (int64_t)blockIdx.x * gpb + threadIdx.x / LPR;
  const int ch4 = (threadIdx.x % LPR) * 4;
  const int64_t start = w * WS;
  if (start >= n || ch4 >= a.D) return;
  const int64_t end = start + WS < n ? start + WS : n;
  const uint32_t lastkey = keys[end - 1];
  if (lastkey == 0 || end >= n || keys[end] != lastkey) return;          // not open to the right
  if (keys[start] == lastkey && start > 0 && keys[start - 1] == lastkey) return;  // did not start here
  const int64_t re = run_end(keys, end, n, lastkey);
  const int L = (int)((re - 1) / WS - w);                                 // windows w+1 .. w+L continue the run
  if (L > LONG_CHAIN) {
    if (ch4 == 0) {
      int slot = atomicAdd(long_count, 1);
      long_list[slot] = make_int2((int)w, L);
    }
    return;
  }
  float4 tot = ld4(plast + w * a.D + ch4);
  for (int j = 1; j <= L; ++j) tot = add4(tot, ld4(pfirst + (w + j) * a.D + ch4));
  st4(out + key_to_row(lastkey, a.Gsh, a.shift) * a.D + ch4, tot);
}

// one block per long chain: its groups sum contiguous sub-ranges of the chain, then the partial
// sums are added in group order (fixed partition => reproducible)
__global__ __launch_bounds__(256) void pull_long_kernel(const PullArgs a, const uint32_t* __restrict__ keys,
                                                        int WS, float* __restrict__ out,
                                                        const float* __restrict__ pfirst,
                                                        const float* __restrict__ plast,
                                                        const int* __restrict__ long_count,
                                                        const int2* __restrict__ long_list) {
  extern __shared__ float sh[];  // [groups][D]
  const int LPR = a.LPRp;
  const int ng = blockDim.x / LPR;
  const int g = threadIdx.x / LPR;
  const int ch4 = (threadIdx.x % LPR) * 4;
  const bool lane_ok = ch4 < a.D;
  const int count = *long_count;
  for (int c = blockIdx.x; c < count; c += gridDim.x) {
    const int2 e = long_list[c];
    const int64_t w = e.x;
    const int L = e.y;
    const int chunk = (L + ng - 1) / ng;
    const int j0 = 1 + g * chunk, j1 = min(L, (g + 1) * chunk);
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane_ok)
      for (int j = j0; j <= j1; ++j) tot = add4(tot, ld4(pfirst + (w + j) * a.D + ch4));
    if (lane_ok) st4(sh + g * a.D + ch4, tot);
    __syncthreads();
    if (g == 0 && lane_ok) {
      float4 t = ld4(plast + w * a.D + ch4);
      for (int q = 0; q < ng; ++q) t = add4(t, ld4(sh + q * a.D + ch4));
      const uint32_t key = keys[(w + 1) * (int64_t)WS - 1];
      st4(out + key_to_row(key, a.Gsh, a.shift) * a.D + ch4, t);
    }
    __syncthreads();
  }
}

int score_launch_pull(PullArgs& a, const uint32_t* keys, const uint32_t* vals, int64_t n, float* out,
                      float* partials, int64_t partial_floats, hipStream_t s) {
  const int WS = 64;
  int LPR = 1;
  while (LPR < a.D / 4) LPR <<= 1;
  if (LPR > 64) return SCORE_E_SHAPE;
  a.LPRp = LPR;
  int64_t nw = cdiv64(n, WS);
  // partials: pfirst [nw][D] | plast [nw][D] | long-chain counter (4 floats) | long list [nw] int2
  if (2 * nw * a.D + 4 + 2 * nw > partial_floats) return SCORE_E_WORKSPACE;
  float* pfirst = partials;
  float* plast = partials + nw * a.D;
  int* long_count = reinterpret_cast<int*>(plast + nw * a.D);
  int2* long_list = reinterpret_cast<int2*>(long_count + 4);
  hipError_t e = hipMemsetAsync(long_count, 0, 16, s);
  if (e != hipSuccess) return (int)e;
  int gpb = 256 / LPR;
  unsigned blocks = (unsigned)cdiv64(nw, gpb);
  hipLaunchKernelGGL(pull_kernel, dim3(blocks), dim3(256), 0, s, a, keys, vals, n, WS, out, pfirst, plast);
  SCORE_CHECK_LAUNCH();
  hipLaunchKernelGGL(pull_fixup_kernel, dim3(blocks), dim3(256), 0, s, a, keys, n, WS, out, pfirst, plast,
                     long_count, long_list);
  SCORE_CHECK_LAUNCH();
  hipLaunchKernelGGL(pull_long_kernel, dim3(1024), dim3(256), (size_t)gpb * a.D * sizeof(float), s, a, keys, WS,
                     out, pfirst, plast, long_count, long_list);
  SCORE_CHECK_LAUNCH();
  return 0;
}
