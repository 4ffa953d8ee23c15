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
#include <rocprim/device/device_scan.hpp>
#include "common.h"
#include "kernels.h"

// occurrence descriptor: seg[31:29] f[28:26] k[25:21] bt[20:0]
#define DESC(seg, f, k, bt) (((uint32_t)(seg) << 29) | ((uint32_t)(f) << 26) | ((uint32_t)(k) << 21) | (uint32_t)(bt))

__global__ void plan_fill_kernel(PlanFillArgs a, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > a.off[6]) return;
  if (i == a.off[6]) {  // sentinel occurrence of the dummy row: unique position 0 is always row 0
    keys[i] = 0;
    vals[i] = DESC(7, 0, 0, 0);
    return;
  }
  int seg = 0;
#pragma unroll
  for (int s = 1; s < 6; ++s) seg += (i >= a.off[s]) ? 1 : 0;
  int64_t local = i - a.off[seg];
  const uint32_t F = (uint32_t)a.F[seg];
  uint32_t f, k, bt;
  // 32-bit index arithmetic (one tensor holds < 2^31 ids: B*T <= 2^21, K <= 32, F <= 8): 64-bit division
  // made this trivial kernel 35 us
  const uint32_t l32 = (uint32_t)local;
  if (seg < 4) {
    const uint32_t q = l32 / F;
    f = l32 - q * F;
    bt = q / (uint32_t)a.K;            // b * TA + t: the occurrence space holds the active slices only
    k = q - bt * (uint32_t)a.K;
    if (a.TA != a.T) {                 // position inside the [B, T, K, F] index tensor
      const uint32_t b = bt / (uint32_t)a.TA;
      local = (((int64_t)b * a.T + (bt - b * (uint32_t)a.TA)) * a.K + k) * F + f;
    }
  } else {
    bt = l32 / F;
    f = l32 - bt * F;
    k = 0;
  }
  uint32_t row = (uint32_t)a.idx[seg][local];
  if (row >= a.n_rows) {     // outside the table (tf.nn.embedding_lookup raises, score.py:51-66): the dummy row, reported
    row = 0;
    if (a.id_status) atomicOr(a.id_status, 1 << ((0x542130 >> (4 * seg)) & 15));   // segment -> position in the feed tuple
  }
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

// keys_out/vals_out <- occurrences of the batch sorted by (owner, row), equal keys in occurrence order.
// which = 0: by size -- sort.hip's six-launch sort below SCORE_OWN_SORT_MAX_N occurrences (the reference's own batch sizes:
// the library takes 19 launches there and the step is bound by the host's launch calls), rocPRIM's onesweep above (cfg-3's
// 2.9 M: the device side decides, and there the library's 8-bit passes coalesce better); 1 / 2 force the library / sort.hip.
// Both sorts are stable: the plan is the same bits either way.
#define SCORE_OWN_SORT_MAX_N (1 << 21)
int score_launch_plan(const PlanFillArgs& a, int key_bits, uint32_t* keys_in, uint32_t* vals_in, uint32_t* keys_out,
                      uint32_t* vals_out, void* temp, size_t temp_bytes, hipStream_t s, int which) {
  int64_t n = a.off[6] + 1;   // + sentinel
  if (which == 2 || (which == 0 && n < SCORE_OWN_SORT_MAX_N))
    return score_launch_plan_own(a, key_bits, keys_in, vals_in, keys_out, vals_out, temp, temp_bytes, s);
  hipLaunchKernelGGL(plan_fill_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, a, keys_in, vals_in);
  SCORE_CHECK_LAUNCH();
  size_t need = 0;
  SCORE_TRY(score_plan_temp_bytes(n, key_bits, &need));
  if (need > temp_bytes) return SCORE_E_WORKSPACE;
  hipError_t e = rocprim::radix_sort_pairs(temp, need, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                           (unsigned)key_bits, s);
  return e == hipSuccess ? 0 : (int)e;
}

// ------------------------------------------------------------------ unique rows / owner offsets / remap
__global__ void plan_flags_kernel(const uint32_t* __restrict__ keys, int64_t n, uint32_t* __restrict__ flags) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// uid[i] = (#run heads in [0, i]) - 1 ; unique_keys[uid] = key of the run ; meta[0] = U
__global__ void plan_unique_kernel(const uint32_t* __restrict__ keys, int64_t n, uint32_t* __restrict__ uid,
                                   uint32_t* __restrict__ unique_keys, int32_t* __restrict__ meta) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t u = uid[i] - 1;   // inclusive scan of the head flags
  uid[i] = u;
  if (i == 0 || keys[i] != keys[i - 1]) unique_keys[u] = keys[i];
  if (i == n - 1) meta[0] = (int32_t)(u + 1);
}
// meta[1 + o] = first unique position whose owner >= o  (o = 0..G) ; unique_rows[u] = local row index
__global__ void plan_offsets_kernel(uint32_t* __restrict__ unique_keys, int32_t* __restrict__ meta, int G, int shift) {
  const int U = meta[0];
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t <= G) {
    uint32_t bound = (t == G) ? 0xFFFFFFFFu : ((uint32_t)t << shift);
    int lo = 0, hi = U;
    if (t == G) lo = U;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (unique_keys[mid] < bound) lo = mid + 1; else hi = mid;
    }
    meta[1 + t] = lo;
  }
}
__global__ void plan_localrow_kernel(const uint32_t* __restrict__ unique_keys, const int32_t* __restrict__ meta,
                                     int shift, int32_t* __restrict__ unique_rows) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= meta[0]) return;
  unique_rows[i] = (int32_t)(unique_keys[i] & ((1u << shift) - 1u));
}
// remapped index tensors: every occurrence -> position of its row in the unique list
__global__ void plan_remap_kernel(PlanRemapArgs a, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ uid,
                                  int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t desc = vals[i];
  const int seg = desc >> 29;
  if (seg > 5) return;
  const int f = (desc >> 26) & 7, k = (desc >> 21) & 31;
  int64_t bt = desc & 0x1FFFFF;
  if (seg < 4 && a.TA != a.T) {   // b * TA + t -> b * T + t
    const int64_t b = bt / a.TA;
    bt = b * a.T + (bt - b * a.TA);
  }
  const int64_t local = seg < 4 ? (bt * a.K + k) * a.F[seg] + f : bt * a.F[seg] + f;
  a.out[seg][local] = (int32_t)uid[i];
}

int score_launch_plan_unique(const PlanRemapArgs& ra, const uint32_t* keys, const uint32_t* vals, int64_t n,
                             uint32_t* flags_scratch, uint32_t* uid, uint32_t* unique_keys, int32_t* unique_rows,
                             int32_t* meta, int G, int shift, void* temp, size_t temp_bytes, hipStream_t s, bool remap) {
  unsigned blocks = (unsigned)cdiv64(n, 256);
  hipLaunchKernelGGL(plan_flags_kernel, dim3(blocks), dim3(256), 0, s, keys, n, flags_scratch);
  SCORE_CHECK_LAUNCH();
  size_t need = 0;
  hipError_t e = rocprim::inclusive_scan(nullptr, need, flags_scratch, uid, (size_t)n, rocprim::plus<uint32_t>(), s);
  if (e != hipSuccess) return (int)e;
  if (need > temp_bytes) return SCORE_E_WORKSPACE;
  e = rocprim::inclusive_scan(temp, need, flags_scratch, uid, (size_t)n, rocprim::plus<uint32_t>(), s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(plan_unique_kernel, dim3(blocks), dim3(256), 0, s, keys, n, uid, unique_keys, meta);
  SCORE_CHECK_LAUNCH();
  hipLaunchKernelGGL(plan_offsets_kernel, dim3(1), dim3(128), 0, s, unique_keys, meta, G, shift);
  SCORE_CHECK_LAUNCH();
  hipLaunchKernelGGL(plan_localrow_kernel, dim3(blocks), dim3(256), 0, s, unique_keys, meta, shift, unique_rows);
  SCORE_CHECK_LAUNCH();
  if (!remap) return 0;          // (only the unique row list was asked for: score_index_plan dedup == 2)
  hipLaunchKernelGGL(plan_remap_kernel, dim3(blocks), dim3(256), 0, s, ra, vals, uid, n);
  SCORE_CHECK_LAUNCH();
  return 0;
}

int score_scan_temp_bytes(int64_t n, size_t* bytes) {
  uint32_t* nul = nullptr;
  hipError_t e = rocprim::inclusive_scan(nullptr, *bytes, nul, nul, (size_t)n, rocprim::plus<uint32_t>(),
                                         (hipStream_t)0);
  return e == hipSuccess ? 0 : (int)e;
}

// ------------------------------------------------------------------ sum rows by destination (owner side)
// out[rows[j]] = sum_j src[j]   over equal rows, in slot order (reproducible).  Used by the shard owner to
// combine the row gradients every rank sent it.
__global__ void rowsum_fill_kernel(const int32_t* __restrict__ rows, int64_t n, uint32_t* __restrict__ keys,
                                   uint32_t* __restrict__ vals) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = (uint32_t)rows[i];
  vals[i] = (6u << 29) | (uint32_t)i;
}

// ------------------------------------------------------------------ pull
// destination row of a run: its unique position (sharded: grads of the mini-table) or the row id itself
__device__ __forceinline__ int64_t out_row(const PullArgs& a, uint32_t key, int64_t idx_in_run) {
  return a.uid ? (int64_t)a.uid[idx_in_run] : (int64_t)key;
}

// final value of a run's row (+ the "written this step" mark score_adam_rows consumes)
__device__ __forceinline__ void store_row(const PullArgs& a, float* __restrict__ out, uint32_t key, int64_t idx_in_run,
                                          int ch4, const float4& v) {
  const int64_t r = out_row(a, key, idx_in_run);
  st4(out + r * a.D + ch4, v);
  if (a.flags && ch4 == 0) a.flags[r] = 2;
}

// Contribution of one occurrence without data-dependent branches, so the loads of several occurrences can be
// in flight together.  MODE 0: owner-side row sum (descriptor = source slot).  MODE 1: model segments with
// constant coefficients (RCA / RRN).  MODE 2: with the co-attention's per-(unit,k) coefficients.  Segments the
// model does not use and the sentinel are clamped onto segment 4/5's pointers; their value is discarded by the
// caller (their key is the dummy row 0).
struct PullSeg {          // per-segment fields of PullArgs, looked up from LDS by the descriptor's segment
  const float* G; const float* cA; const float* cB; const float* Wv;
  int ldg, gcol; float constA; int useA;
};
template <int MODE>
__device__ __forceinline__ float4 pull_contrib_t(const PullArgs& a, const PullSeg* __restrict__ tab, uint32_t desc,
                                                 int ch4) {
  if (MODE == 0) return ld4(a.G[0] + (int64_t)(desc & 0x1FFFFFFF) * a.D + ch4);
  int seg = desc >> 29;
  seg = seg > 5 ? 5 : seg;
  const int f = (desc >> 26) & 7, k = (desc >> 21) & 31;
  const int64_t bt = desc & 0x1FFFFF;
  const int col = f * a.D + ch4;
  const PullSeg si = tab[seg];               // ds_read: no branch, so the loads below batch across occurrences
  const float4 g = ld4_global(si.G + bt * si.ldg + si.gcol + col);
  float ca = si.constA;
  float4 r;
  if (MODE == 2) {
    // every segment has valid cA / cB / Wv pointers (the targets' point at the first call's arrays): the
    // loads are unconditional, useA / constA decide what is used
    const float pa = ld1_global(si.cA + bt * a.K + k);
    const float cb = ld1_global(si.cB + bt * a.K + k);
    const float4 wv = ld4_global(si.Wv + col);
    ca = (si.useA & 1) ? pa : ca;
    r = make_float4(ca * g.x, ca * g.y, ca * g.z, ca * g.w);
    r = fma4((si.useA & 2) ? cb : 0.f, wv, r);
  } else {
    r = make_float4(ca * g.x, ca * g.y, ca * g.z, ca * g.w);
  }
  return r;
}

// U occurrences per trip: their keys, descriptors and contributions are loaded before the run logic consumes
// them in order (the walk used to be one dependent global-memory latency per occurrence)
#define PULL_U 8
template <int MODE>
__global__ __launch_bounds__(256) void pull_kernel(const PullArgs a, const uint32_t* __restrict__ keys,
                                                   const uint32_t* __restrict__ vals, int64_t n, int WS,
                                                   float* __restrict__ out, float* __restrict__ pfirst,
                                                   float* __restrict__ plast) {
  const int LPR = a.LPRp;                                // lanes per group (power of two >= D/4)
  const int gpb = blockDim.x / LPR;                      // groups per block
  const int64_t w = (int64_t)blockIdx.x * gpb + threadIdx.x / LPR;
  const int ch4 = (threadIdx.x % LPR) * 4;
  __shared__ PullSeg tab[6];
  if (MODE != 0) {
    if (threadIdx.x == 0) {
#pragma unroll
      for (int sg = 0; sg < 6; ++sg) {
        PullSeg t;
        t.G = a.G[sg] ? a.G[sg] : a.G[4];
        t.ldg = a.ldg[sg]; t.gcol = a.gcol[sg]; t.constA = a.constA[sg];
        const int c = (sg & 2) ? 2 : 0;                       // segments 0,1 -> call 0; 2,3 -> call 1; targets -> call 0
        t.cA = a.cA[sg < 4 ? c : 0]; t.cB = a.cB[sg < 4 ? c : 0];
        t.Wv = a.Wv[sg < 4 ? sg : sg - 4];
        t.useA = (sg < 4 && a.cA[sg] ? 1 : 0) | (sg < 4 && a.cB[sg] ? 2 : 0);
        tab[sg] = t;
      }
    }
    __syncthreads();
  }
  const int64_t start = w * WS;
  if (start >= n || ch4 >= a.D) return;
  const int64_t end = start + WS < n ? start + WS : n;
  if (a.zero_is_dummy && keys[end - 1] == 0) return;    // keys ascend: a window of dummy-row uses only
  uint32_t cur = keys[start];
  // destination of the current run: the row id itself, or (sharded plan) its unique position -- the same for every
  // occurrence of a run, so it is read with the trip's other loads and carried along (a lookup at each store
  // had put a vmcnt(0) wait, i.e. a drain of everything in flight, behind every row: 0.27 -> 0.40 ms sharded)
  const bool has_uid = a.uid != nullptr;
  uint32_t cur_dst = has_uid ? a.uid[start] : cur;
  const bool first_open = start > 0 && keys[start - 1] == cur;
  bool is_first = true;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto store_cur = [&]() {
    st4(out + (int64_t)cur_dst * a.D + ch4, acc);
    if (a.flags && ch4 == 0) a.flags[cur_dst] = 2;
  };
  for (int64_t i0 = start; i0 < end; i0 += PULL_U) {
    uint32_t kk[PULL_U], vv[PULL_U], uu[PULL_U];
    float4 cc[PULL_U];
#pragma unroll
    for (int u = 0; u < PULL_U; ++u) {
      const int64_t idx = i0 + u < end ? i0 + u : end - 1;
      kk[u] = keys[idx];
      vv[u] = vals[idx];
    }
    if (has_uid) {
#pragma unroll
      for (int u = 0; u < PULL_U; ++u) uu[u] = a.uid[i0 + u < end ? i0 + u : end - 1];
    } else {
#pragma unroll
      for (int u = 0; u < PULL_U; ++u) uu[u] = kk[u];
    }
#pragma unroll
    for (int u = 0; u < PULL_U; ++u) cc[u] = pull_contrib_t<MODE>(a, tab, vv[u], ch4);
#pragma unroll
    for (int u = 0; u < PULL_U; ++u) {
      if (i0 + u >= end) break;
      const uint32_t key = kk[u];
      if (key != cur) {
        if (cur != 0 || !a.zero_is_dummy) {
          if (is_first && first_open) st4(pfirst + w * a.D + ch4, acc);
          else store_cur();
        }
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        cur = key;
        cur_dst = uu[u];
        is_first = false;
      }
      if (key != 0 || !a.zero_is_dummy) acc = add4(acc, cc[u]);
    }
  }
  if (cur != 0 || !a.zero_is_dummy) {
    const bool open_right = end < n && keys[end] == cur;
    if (is_first && first_open) st4(pfirst + w * a.D + ch4, acc);
    else if (open_right) st4(plast + w * a.D + ch4, acc);
    else store_cur();
  }
}

// first index in [lo, n) whose key differs from `key` (keys ascending, keys[lo-1] == key).  Most runs end within a
// window or two: gallop (lo+1, +2, +4, ...) to bracket the end, then bisect the bracket -- a plain bisection of
// [lo, n) is ~21 dependent loads for every open window.
__device__ __forceinline__ int64_t run_end(const uint32_t* __restrict__ keys, int64_t lo, int64_t n, uint32_t key) {
  int64_t step = 1, hi = n;
  while (lo + step < n) {
    if (keys[lo + step - 1] != key) { hi = lo + step - 1; break; }
    lo += step;
    step <<= 1;
  }
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (keys[mid] == key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// A run that starts in window w and continues to the right: total = plast[w] + pfirst[w+1] + ...
// (window order, so the sum is reproducible).  Chains longer than LONG_CHAIN windows (the hot categorical rows:
// thousands of windows) are taken by the WHOLE workgroup afterwards: its groups sum contiguous sub-ranges of the chain,
// then the partial sums are added in group order (fixed partition => reproducible).  (Until round 4 those chains were
// queued for a third launch, one workgroup per chain there as here: 6 - 11 us and a dependent launch on the step's chain.)
#define LONG_CHAIN 16
__global__ __launch_bounds__(256) void pull_fixup_kernel(const PullArgs a, const uint32_t* __restrict__ keys,
                                                         int64_t n, int WS, float* __restrict__ out,
                                                         const float* __restrict__ pfirst,
                                                         const float* __restrict__ plast) {
  extern __shared__ float sh[];              // [groups][D]: a long chain's partial sums
  __shared__ int2 s_long[256];               // (window, L) of the long chains that start in this workgroup's windows
  __shared__ int s_nlong;
  const int LPR = a.LPRp;
  const int gpb = blockDim.x / LPR;
  const int g = threadIdx.x / LPR;
  const int64_t w = (int64_t)blockIdx.x * gpb + g;
  const int ch4 = (threadIdx.x % LPR) * 4;
  const bool lane_ok = ch4 < a.D;
  if (threadIdx.x == 0) s_nlong = 0;
  __syncthreads();
  const int64_t start = w * WS;
  if (start < n && lane_ok) {
    const int64_t end = start + WS < n ? start + WS : n;
    const uint32_t lastkey = keys[end - 1];
    const bool open_right = !((lastkey == 0 && a.zero_is_dummy) || end >= n || keys[end] != lastkey);
    const bool starts_here = !(keys[start] == lastkey && start > 0 && keys[start - 1] == lastkey);
    if (open_right && starts_here) {
      const int64_t re = run_end(keys, end, n, lastkey);
      const int L = (int)((re - 1) / WS - w);                               // windows w+1 .. w+L continue the run
      if (L > LONG_CHAIN) {
        if (ch4 == 0) s_long[atomicAdd(&s_nlong, 1)] = make_int2(g, L);     // (the order of the list changes no sum)
      } else {
        float4 tot = ld4(plast + w * a.D + ch4);
        for (int j = 1; j <= L; ++j) tot = add4(tot, ld4(pfirst + (w + j) * a.D + ch4));
        store_row(a, out, lastkey, end - 1, ch4, tot);
      }
    }
  }
  __syncthreads();
  const int count = s_nlong;
  for (int c = 0; c < count; ++c) {
    const int2 e = s_long[c];
    const int64_t wl = (int64_t)blockIdx.x * gpb + e.x;
    const int L = e.y;
    const int chunk = (L + gpb - 1) / gpb;
    const int j0 = 1 + g * chunk, j1 = min(L, (g + 1) * chunk);
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane_ok) {
      // eight partial sums: eight loads in flight per trip (one dependent add chain waits a full memory latency per
      // window); fixed association, so still reproducible
      float4 t8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t8[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      int j = j0;
      for (; j + 7 <= j1; j += 8) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = ld4(pfirst + (wl + j + q) * a.D + ch4);
#pragma unroll
        for (int q = 0; q < 8; ++q) t8[q] = add4(t8[q], v[q]);
      }
      for (; j <= j1; ++j) t8[0] = add4(t8[0], ld4(pfirst + (wl + j) * a.D + ch4));
      tot = add4(add4(add4(t8[0], t8[1]), add4(t8[2], t8[3])), add4(add4(t8[4], t8[5]), add4(t8[6], t8[7])));
      st4(sh + g * a.D + ch4, tot);
    }
    __syncthreads();
    if (g == 0 && lane_ok) {
      float4 t = ld4(plast + wl * a.D + ch4);
      for (int q = 0; q < gpb; ++q) t = add4(t, ld4(sh + q * a.D + ch4));
      const uint32_t key = keys[(wl + 1) * (int64_t)WS - 1];
      store_row(a, out, key, (wl + 1) * (int64_t)WS - 1, ch4, t);
    }
    __syncthreads();
  }
}

// occurrences per window: 64 for a whole batch (millions of occurrences), fewer for short lists so the
// chip still gets >= ~16k independent groups
int score_pull_window(int64_t n) { return n >= (1 << 20) ? 64 : n >= (1 << 19) ? 32 : n >= (1 << 18) ? 16 : 8; }

int score_launch_pull(PullArgs& a, const uint32_t* keys, const uint32_t* vals, int64_t n, float* out,
                      float* partials, int64_t partial_floats, hipStream_t s) {
  const int WS = score_pull_window(n);
  int LPR = 1;
  while (LPR < a.D / 4) LPR <<= 1;
  if (LPR > 64) return SCORE_E_SHAPE;
  a.LPRp = LPR;
  int64_t nw = cdiv64(n, WS);
  // partials: pfirst [nw][D] | plast [nw][D]
  if (2 * nw * a.D > partial_floats) return SCORE_E_WORKSPACE;
  float* pfirst = partials;
  float* plast = partials + nw * a.D;
  int gpb = 256 / LPR;
  unsigned blocks = (unsigned)cdiv64(nw, gpb);
  // contribution form: owner-side row sum (descriptors are source slots), constant coefficients, or the
  // co-attention's per-(unit, k) coefficients
  const int mode = (a.G[1] == nullptr && a.ldg[0] == 0) ? 0 : (a.cA[0] ? 2 : 1);
  if (mode == 0) hipLaunchKernelGGL(pull_kernel<0>, dim3(blocks), dim3(256), 0, s, a, keys, vals, n, WS, out, pfirst, plast);
  else if (mode == 1) hipLaunchKernelGGL(pull_kernel<1>, dim3(blocks), dim3(256), 0, s, a, keys, vals, n, WS, out, pfirst, plast);
  else hipLaunchKernelGGL(pull_kernel<2>, dim3(blocks), dim3(256), 0, s, a, keys, vals, n, WS, out, pfirst, plast);
  SCORE_CHECK_LAUNCH();
  hipLaunchKernelGGL(pull_fixup_kernel, dim3(blocks), dim3(256), (size_t)gpb * a.D * sizeof(float), s, a, keys, n, WS, out,
                     pfirst, plast);
  SCORE_CHECK_LAUNCH();
  return 0;
}

int score_rowsum_temp_bytes(int64_t n, size_t* bytes) { return score_plan_temp_bytes(n, 32, bytes); }

extern "C" int score_segment_sum_rows(const int32_t* rows, const float* src, int64_t n, int32_t D, int64_t n_out_rows,
                                      float* out, uint8_t* row_flags, void* scratch, int64_t scratch_bytes,
                                      void* stream) {
  if (!rows || !src || !out || !scratch || n < 0 || D <= 0 || (D & 3) || n_out_rows <= 0) return SCORE_E_BADARG;
  if (n == 0) return 0;
  if (n >= (1 << 29)) return SCORE_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  size_t sort_bytes = 0;
  SCORE_TRY(score_plan_temp_bytes(n, 32, &sort_bytes));
  const int64_t nw = cdiv64(n, score_pull_window(n));
  const int64_t partial_floats = 2 * nw * D + 8 + 2 * nw;
  // scratch: keys_in | keys_out | vals_in | vals_out | partials | sort temp
  const int64_t need = 4 * align_up64(n, 4) * 4 + align_up64(partial_floats, 4) * 4 + (int64_t)sort_bytes;
  if (need > scratch_bytes) return SCORE_E_WORKSPACE;
  uint32_t* keys_in = static_cast<uint32_t*>(scratch);
  uint32_t* keys_out = keys_in + align_up64(n, 4);
  uint32_t* vals_in = keys_out + align_up64(n, 4);
  uint32_t* vals_out = vals_in + align_up64(n, 4);
  float* partials = reinterpret_cast<float*>(vals_out + align_up64(n, 4));
  void* temp = partials + align_up64(partial_floats, 4);
  hipLaunchKernelGGL(rowsum_fill_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, rows, n, keys_in, vals_in);
  SCORE_CHECK_LAUNCH();
  int key_bits = 1;
  while (key_bits < 32 && ((int64_t)1 << key_bits) < n_out_rows) ++key_bits;
  hipError_t e = rocprim::radix_sort_pairs(temp, sort_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                           (unsigned)key_bits, s);
  if (e != hipSuccess) return (int)e;
  PullArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.D = D; pa.K = 1; pa.G[0] = src; pa.zero_is_dummy = 0; pa.flags = row_flags;   // local row 0 is a real row on shards > 0
  return score_launch_pull(pa, keys_out, vals_out, n, out, partials, partial_floats, s);
}

// out[rows[i]] (+)= src[i] for one source's row list (rows unique inside the call): the first writer of a
// row this step stores, later ones add -- called once per source rank in rank order, the sum is reproducible
// without a sort or an atomic.  Marks every row it touches (state 2, see score_adam_rows).
__global__ __launch_bounds__(256) void rows_accumulate_kernel(const int32_t* __restrict__ rows,
                                                              const float* __restrict__ src, int64_t n, int D,
                                                              int LPR, float* __restrict__ out,
                                                              uint8_t* __restrict__ flags) {
  const int gpb = blockDim.x / LPR;
  const int64_t i = (int64_t)blockIdx.x * gpb + threadIdx.x / LPR;
  const int ch4 = (threadIdx.x % LPR) * 4;
  if (i >= n || ch4 >= D) return;
  const int64_t r = rows[i];
  const uint8_t f = flags[r];
  float4 v = ld4(src + i * D + ch4);
  if (f == 2) v = add4(ld4(out + r * D + ch4), v);
  st4(out + r * D + ch4, v);
  if (ch4 == 0 && f != 2) flags[r] = 2;   // (the group's lanes sit in one wave: all of them read f above)
}

extern "C" int score_rows_accumulate(const int32_t* rows, const float* src, int64_t n, int32_t D, int64_t n_out_rows,
                                     float* out, uint8_t* row_flags, void* stream) {
  if (!rows || !src || !out || !row_flags || n < 0 || D <= 0 || (D & 3) || D > 256 || n_out_rows <= 0)
    return SCORE_E_BADARG;
  if (n == 0) return 0;
  int LPR = 1;
  while (LPR < D / 4) LPR <<= 1;
  const int gpb = 256 / LPR;
  hipLaunchKernelGGL(rows_accumulate_kernel, dim3((unsigned)cdiv64(n, gpb)), dim3(256), 0, (hipStream_t)stream, rows,
                     src, n, D, LPR, out, row_flags);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// The same for ALL source ranks in one launch.  rows = the sources' lists back to back (source p: [off[p], off[p+1]),
// each list unique and ASCENDING -- they are segments of score_index_plan's unique-row lists).  The slot of the LOWEST
// source that names a row owns the row: it adds the later sources' contributions in source order (a binary search per
// later list) and stores once -- the association ((g_p0 + g_p1) + g_p2) ... of the per-source launches, so the result is
// the same bits; slots whose row an earlier source names do nothing.  Eight ranks: one launch instead of eight.
#define ACC_MAX_SOURCES 64
struct AccMultiArgs { int64_t off[ACC_MAX_SOURCES + 1]; int n_sources; };
__device__ __forceinline__ int64_t acc_find(const int32_t* __restrict__ rows, int64_t lo, int64_t hi, int32_t r) {
  while (lo < hi) {                        // first index in [lo, hi) with rows[idx] >= r
    const int64_t mid = (lo + hi) >> 1;
    if (rows[mid] < r) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__global__ __launch_bounds__(256) void rows_accumulate_multi_kernel(const AccMultiArgs a, const int32_t* __restrict__ rows,
                                                                    const float* __restrict__ src, int D, int LPR,
                                                                    float* __restrict__ out, uint8_t* __restrict__ flags) {
  const int gpb = blockDim.x / LPR;
  const int64_t i = (int64_t)blockIdx.x * gpb + threadIdx.x / LPR;
  const int ch4 = (threadIdx.x % LPR) * 4;
  const int64_t n = a.off[a.n_sources];
  if (i >= n) return;
  int p = 0;
  while (p + 1 < a.n_sources && i >= a.off[p + 1]) ++p;
  const int32_t r = rows[i];
  for (int q = 0; q < p; ++q) {            // an earlier source names the row: that slot sums it
    const int64_t j = acc_find(rows, a.off[q], a.off[q + 1], r);
    if (j < a.off[q + 1] && rows[j] == r) return;
  }
  if (ch4 >= D) return;
  float4 v = ld4(src + i * D + ch4);
  for (int q = p + 1; q < a.n_sources; ++q) {
    const int64_t j = acc_find(rows, a.off[q], a.off[q + 1], r);
    if (j < a.off[q + 1] && rows[j] == r) v = add4(v, ld4(src + j * D + ch4));
  }
  const uint8_t f = flags[r];
  if (f == 2) v = add4(ld4(out + (int64_t)r * D + ch4), v);      // (a row already written this step by an earlier call)
  st4(out + (int64_t)r * D + ch4, v);
  if (ch4 == 0 && f != 2) flags[r] = 2;
}

extern "C" int score_rows_accumulate_multi(const int32_t* rows, const float* src, const int64_t* offsets, int32_t n_sources,
                                           int32_t D, int64_t n_out_rows, float* out, uint8_t* row_flags, void* stream) {
  if (!rows || !src || !offsets || !out || !row_flags || n_sources < 1 || n_sources > ACC_MAX_SOURCES || D <= 0 || (D & 3) ||
      D > 256 || n_out_rows <= 0)
    return SCORE_E_BADARG;
  AccMultiArgs a;
  for (int p = 0; p <= n_sources; ++p) {
    a.off[p] = offsets[p];
    if (p && offsets[p] < offsets[p - 1]) return SCORE_E_BADARG;
  }
  if (a.off[0] != 0) return SCORE_E_BADARG;
  a.n_sources = n_sources;
  const int64_t n = a.off[n_sources];
  if (n == 0) return 0;
  int LPR = 1;
  while (LPR < D / 4) LPR <<= 1;
  const int gpb = 256 / LPR;
  hipLaunchKernelGGL(rows_accumulate_multi_kernel, dim3((unsigned)cdiv64(n, gpb)), dim3(256), 0, (hipStream_t)stream, a, rows,
                     src, D, LPR, out, row_flags);
  SCORE_CHECK_LAUNCH();
  return 0;
}

extern "C" int64_t score_segment_sum_scratch_bytes(int64_t n, int32_t D) {
  size_t sort_bytes = 0;
  if (score_plan_temp_bytes(n > 0 ? n : 1, 32, &sort_bytes) != 0) return -1;
  const int64_t nw = cdiv64(n, score_pull_window(n));
  const int64_t partial_floats = 2 * nw * D + 8 + 2 * nw;
  return 4 * align_up64(n, 4) * 4 + align_up64(partial_floats, 4) * 4 + (int64_t)sort_bytes + 64;
}

// ------------------------------------------------------------------ AUC / log-loss of an evaluation pass
__global__ void auc_keys_kernel(const float* __restrict__ pred, const int32_t* __restrict__ label, int64_t n,
                                uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = __float_as_uint(pred[i]);     // scores are positive floats: their bit patterns sort like the values
  vals[i] = label[i] != 0;
}
// per block: sum over its positives of the average rank of their tie group, the positives' count, and the
// log-loss terms; groups are found by binary search in the sorted keys (robust to long runs of equal scores)
__global__ __launch_bounds__(256) void auc_partials_kernel(const uint32_t* __restrict__ keys,
                                                           const uint32_t* __restrict__ lab, int64_t n,
                                                           double* __restrict__ part) {
  __shared__ double sh[3][256];
  double rk = 0.0, np = 0.0, ll = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t k = keys[i];
    const bool pos = lab[i] != 0;
    if (pos) {
      int64_t lo = 0, hi = i;                 // first index with key == k
      while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (keys[mid] < k) lo = mid + 1; else hi = mid; }
      const int64_t first = lo;
      lo = i; hi = n;                         // first index with key > k
      while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (keys[mid] <= k) lo = mid + 1; else hi = mid; }
      rk += 0.5 * (double)(first + 1 + lo);   // average of the 1-based ranks first+1 .. lo
      np += 1.0;
    }
    const double eps = 2.220446049250313e-16;
    double p = (double)__uint_as_float(k);
    p = p < eps ? eps : (p > 1.0 - eps ? 1.0 - eps : p);
    ll -= pos ? log(p) : log(1.0 - p);
  }
  sh[0][threadIdx.x] = rk; sh[1][threadIdx.x] = np; sh[2][threadIdx.x] = ll;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int q = 0; q < 3; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int q = 0; q < 3; ++q) part[q * gridDim.x + blockIdx.x] = sh[q][0];
}
__global__ void auc_final_kernel(const double* __restrict__ part, int nparts, int64_t n, double* __restrict__ out2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double rk = 0.0, np = 0.0, ll = 0.0;
  for (int i = 0; i < nparts; ++i) { rk += part[i]; np += part[nparts + i]; ll += part[2 * nparts + i]; }
  const double nn = (double)n - np;
  out2[0] = (np > 0.0 && nn > 0.0) ? (rk - np * (np + 1.0) * 0.5) / (np * nn) : nan("");
  out2[1] = ll / (double)n;
}

#define AUC_PARTS 256
extern "C" int64_t score_auc_scratch_bytes(int64_t n) {
  size_t sort_bytes = 0;
  if (n <= 0 || score_plan_temp_bytes(n, 32, &sort_bytes) != 0) return -1;
  return 4 * align_up64(n, 4) * 4 + 3 * AUC_PARTS * 8 + (int64_t)sort_bytes + 64;
}

extern "C" int score_auc_logloss(const float* pred, const int32_t* label, int64_t n, double* out2, void* scratch,
                                 int64_t scratch_bytes, void* stream) {
  if (!pred || !label || !out2 || !scratch || n <= 0) return SCORE_E_BADARG;
  if (n >= (1ll << 31)) return SCORE_E_SHAPE;
  if (scratch_bytes < score_auc_scratch_bytes(n)) return SCORE_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  size_t sort_bytes = 0;
  SCORE_TRY(score_plan_temp_bytes(n, 32, &sort_bytes));
  const int64_t n4 = align_up64(n, 4);
  uint32_t* keys_in = static_cast<uint32_t*>(scratch);
  uint32_t* keys_out = keys_in + n4;
  uint32_t* vals_in = keys_out + n4;
  uint32_t* vals_out = vals_in + n4;
  double* part = reinterpret_cast<double*>(vals_out + n4);
  void* temp = part + 3 * AUC_PARTS;
  hipLaunchKernelGGL(auc_keys_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, pred, label, n, keys_in, vals_in);
  SCORE_CHECK_LAUNCH();
  hipError_t e = rocprim::radix_sort_pairs(temp, sort_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, 32u, s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(auc_partials_kernel, dim3(AUC_PARTS), dim3(256), 0, s, keys_out, vals_out, n, part);
  SCORE_CHECK_LAUNCH();
  hipLaunchKernelGGL(auc_final_kernel, dim3(1), dim3(64), 0, s, part, AUC_PARTS, n, out2);
  SCORE_CHECK_LAUNCH();
  return 0;
}
