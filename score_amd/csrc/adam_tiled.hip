// Time-tiled ApplyAdam over the embedding table (score.py:96-99 applied to emb_mtx, dense).
//
// tf.train.AdamOptimizer on the masked dense table moves EVERY row each step: a row no sample of the batch used
// still gets m *= beta1, v *= beta2, p -= alpha_t m / (sqrt(v) + eps).  In steady state that sweep is six fp32
// streams over the whole table (2.35 GB at cfg-3, 0.48 ms of a 1.78 ms step) although a batch touches ~12 % of
// the rows.  The zero-gradient update of a row reads nothing but the row itself and the step's alpha, so it can be
// applied later, in step order, the first time anybody needs the row -- the same fp32 operations in the same
// order, hence the same bits.  Per row we keep the number of optimizer steps already applied (row_step) and per
// step its alpha (a ring of the last SCORE_ADAM_RING values); then
//   score_adam_catchup_ids   before the forward: every row the batch is about to read is brought up to step n-1
//   score_adam_touched       after the backward: rows with a gradient (state 2) get step n from g, like the sweep
//   score_adam_catchup_rows  one 1/window slice of the table per step is brought up to date beside the forward,
//                            so no row ever lags more than `window` steps; over the whole table it is the flush
//                            that every observer of the table runs first (save, get_params, a dense step, ...)
// A row's HBM traffic drops from once per step to once per use (plus once per window).  The arithmetic is not
// skipped: each (row, step) update is executed exactly once, by whichever of the three gets to the row first.
#include <cstdlib>
#include "common.h"
#include "kernels.h"

struct TiledArgs {
  float* p; float* m; float* v; const float* g;
  uint8_t* flags; uint32_t* step; float* ring;
  int64_t n_rows; int D; int LPR;
  float omb1, omb2, eps;
  const int32_t* guard;        // score_adam_table_t.id_status: non-zero -> nothing is applied or replayed
  const int32_t* skipped;      // score_adam_table_t.skipped_steps: optimizer steps suppressed so far (the caller's step count runs ahead by it)
};
__device__ __forceinline__ bool tiled_guarded(const TiledArgs& a) { return a.guard && *a.guard; }

// Publishing a row's ApplyAdam of step `upto` (state 2 -> 1) while OTHER kernels run beside this one (round 5: the window
// slice and the look-ahead catch-up of the next batch's rows on the side stream no longer wait for / hold up the touched-row
// update): those kernels take a row for lagging when they read state 1 and row_step < upto, and must never see the new state
// with the old count (they would replay step `upto` on a row that already has it, with a zero gradient).  The count is
// written first, through to memory (agent scope), and the state byte only when that store has been acknowledged; readers
// load the count past their caches (tiled_row_step): a fresh state 1 then always comes with the fresh count, a stale state
// byte reads 2 (the row looks busy: skipped).  Concurrent kernels have no kernel boundary between them to make plain
// loads coherent across the XCDs' L2s.
__device__ __forceinline__ void tiled_publish_applied(const TiledArgs& a, int64_t row, uint32_t upto) {
  __hip_atomic_store(&a.step[row], upto, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  a.flags[row] = 1;
}
__device__ __forceinline__ uint32_t tiled_row_step(const TiledArgs& a, int64_t row) {
  return __hip_atomic_load(&a.step[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int tiled_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Sub-group `sg` of the wave takes the sg-th set bit of `mask` (a lane index), and the first `nper` set bits
// leave the mask.  `mask` is wave-uniform, so is the loop.
__device__ __forceinline__ int tiled_pick(uint64_t& mask, int sg, int nper) {
  int src = -1;
  for (int j = 0; j < nper && mask; ++j) {
    const int b = __ffsll((long long)mask) - 1;
    if (j == sg) src = b;
    mask &= mask - 1;
  }
  return src;
}
__device__ __forceinline__ uint32_t tiled_wave_min(uint32_t x) {
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t y = (uint32_t)__shfl_xor((int)x, off, SCORE_WAVE);
    x = y < x ? y : x;
  }
  return x;
}

// The lanes whose bit is set in `mask` each hold a row (my_row) to process; groups of LPR lanes take them over, up to
// 64/LPR rows at a time.  MODE 0: one ApplyAdam with the row's gradient (step `upto`, alpha_now).  MODE 1 / 2: the
// zero-gradient updates of steps my_old+1 .. upto; areg of lane L holds alpha of step upto - L (2: state 3 -> 1).
template <int MODE>
__device__ __forceinline__ void tiled_rows(const TiledArgs& a, uint64_t mask, int my_row, uint32_t my_old, uint32_t upto,
                                           float alpha_now, float areg, int lane) {
  const int nper = SCORE_WAVE / a.LPR, sg = lane / a.LPR, ch4 = (lane % a.LPR) * 4;
  int64_t pend = -1;            // MODE 0: row whose count is on its way to memory; its state byte follows (tiled_publish_applied)
  while (mask) {
    const int src = tiled_pick(mask, sg, nper);
    const int srcl = src < 0 ? 0 : src;
    const int row = __shfl(my_row, srcl, SCORE_WAVE);
    const uint32_t old = (uint32_t)__shfl((int)my_old, srcl, SCORE_WAVE);
    const bool on = src >= 0 && ch4 < a.D;
    const int64_t e = (int64_t)row * a.D + ch4;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), m = p, v = p, g = p;
    if (on) {
      p = ld4(a.p + e); m = ld4(a.m + e); v = ld4(a.v + e);
      if (MODE == 0) g = ld4(a.g + e);
    }
    if (MODE == 0) {
      if (on) {
        score_adam1(p.x, m.x, v.x, g.x, a.omb1, a.omb2, alpha_now, a.eps);
        score_adam1(p.y, m.y, v.y, g.y, a.omb1, a.omb2, alpha_now, a.eps);
        score_adam1(p.z, m.z, v.z, g.z, a.omb1, a.omb2, alpha_now, a.eps);
        score_adam1(p.w, m.w, v.w, g.w, a.omb1, a.omb2, alpha_now, a.eps);
        // (the previous trip's rows: this trip's loads have returned, so -- memory operations of a wave complete in order --
        //  has the count's store; the wait below is free)
        if (pend >= 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); a.flags[pend] = 1; pend = -1; }
        st4(a.p + e, p); st4(a.m + e, m); st4(a.v + e, v);
        if (ch4 == 0) {
#if defined(TILED_PROBE_PLAIN_PUBLISH)      // timing probe (tools/build_variant.py): count and state byte as two plain stores, as before round 5
          a.step[row] = upto; a.flags[row] = 1;
#else
          __hip_atomic_store(&a.step[row], upto, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          pend = row;
#endif
        }
      }
    } else {
      const uint32_t smin = (uint32_t)__builtin_amdgcn_readfirstlane((int)tiled_wave_min(on ? old : upto));
      // a lag beyond the ring cannot be replayed: raise the sticky error word behind the ring (the window sweep makes
      // this unreachable; score_adam_catchup_rows' caller checks the word whenever it synchronises anyway)
      if (upto - smin > SCORE_ADAM_RING - 1 && lane == 0) atomicOr(reinterpret_cast<unsigned int*>(a.ring + SCORE_ADAM_RING), 1u);
      for (uint32_t s = smin + 1; s <= upto; ++s) {          // wave-uniform trip count: alpha comes from a lane read
        const float al = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(areg), (int)(upto - s)));
        if (on && s > old) {
          score_adam1(p.x, m.x, v.x, 0.f, a.omb1, a.omb2, al, a.eps);
          score_adam1(p.y, m.y, v.y, 0.f, a.omb1, a.omb2, al, a.eps);
          score_adam1(p.z, m.z, v.z, 0.f, a.omb1, a.omb2, al, a.eps);
          score_adam1(p.w, m.w, v.w, 0.f, a.omb1, a.omb2, al, a.eps);
        }
      }
      if (on) {
        st4(a.p + e, p); st4(a.m + e, m); st4(a.v + e, v);
        if (ch4 == 0) {
          a.step[row] = upto;
          if (MODE == 2) a.flags[row] = 1;
        }
      }
    }
  }
  if (MODE == 0 && pend >= 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); a.flags[pend] = 1; }
}

// The scans of the state bytes: SCAN_V consecutive rows' bytes per lane in one load (bytes beyond n_rows read as 0).
// (SCAN_V: state bytes per lane and trip; 1 is what runs -- see tiled_scan_v)
template <int SCAN_V>
__device__ __forceinline__ uint32_t tiled_load_states(const uint8_t* __restrict__ flags, int64_t r0, int64_t n_rows) {
  if (SCAN_V == 1) return r0 < n_rows ? (uint32_t)flags[r0] : 0u;
  if (r0 + SCAN_V <= n_rows && (reinterpret_cast<uintptr_t>(flags + r0) & 3) == 0)      // (r0 is a multiple of SCAN_V: aligned whenever the array is)
    return *reinterpret_cast<const uint32_t*>(flags + r0);
  uint32_t f = 0;
  for (int j = 0; j < SCAN_V; ++j)
    if (r0 + j < n_rows) f |= (uint32_t)flags[r0 + j] << (8 * j);
  return f;
}
// does any byte of f equal `state` (1 .. 3)?
__device__ __forceinline__ bool tiled_has_state(uint32_t f, uint32_t state) {
  const uint32_t x = f ^ (state * 0x01010101u);
  return ((x - 0x01010101u) & ~x & 0x80808080u) != 0u;
}

// rows whose state byte is 2: the step's ApplyAdam from their gradient.  One lane per row scans the state bytes.
// A state-2 row whose gradient is NOT applied goes back to state 1.  Its row_step: a row that was live before the pass keeps
// its count (it was brought up to date before the forward, or -- in a step queued behind a suppressed one -- still owes what
// it owed); a row that was in state 0 has never had one, and m = v = 0 there, so every update it "owes" is the identity and
// any count is right for it: it gets `applied`, the number of steps really applied so far, which keeps it inside the ring.
// The two cannot be told apart by the state byte, but by the moments they can: all-zero m and v <=> the count is free.
__device__ __forceinline__ void tiled_unmark_row(const TiledArgs& a, int64_t r, uint32_t applied) {
  bool zero = true;
  const float* m = a.m + r * a.D;
  const float* v = a.v + r * a.D;
  for (int c = 0; c < a.D && zero; ++c) zero = m[c] == 0.f && v[c] == 0.f;
  a.flags[r] = 1;
  if (zero) a.step[r] = applied;
}
// steps the device has really applied when the caller says `step` is being applied: the caller's count runs ahead by the
// steps suppressed so far (score_guard_t.skipped; error path only)
__device__ __forceinline__ uint32_t tiled_applied(const TiledArgs& a, uint32_t step) {
  const uint32_t sk = a.skipped ? (uint32_t)*a.skipped : 0u;
  return step - 1 > sk ? step - 1 - sk : 0u;
}
// `drop` (score_adam_unmark) or a set guard word: nothing is applied (tiled_unmark_row).  Virtual block blk of nblk.
template <int SCAN_V>
__device__ __forceinline__ void adam_touched_body(const TiledArgs& a, uint32_t step, float alpha, int drop, int blk, int nblk) {
  const int lane = tiled_lane();
  const int64_t stride = (int64_t)nblk * blockDim.x;
  if (drop || tiled_guarded(a)) {
    const uint32_t applied = drop ? step - 1 : tiled_applied(a, step);
    for (int64_t r = (int64_t)blk * blockDim.x + threadIdx.x; r < a.n_rows; r += stride)
      if (a.flags[r] == 2) tiled_unmark_row(a, r, applied);
    return;
  }
  if (blk == 0 && threadIdx.x == 0) a.ring[step % SCORE_ADAM_RING] = alpha;
  // SCAN_V state bytes per lane and trip
  for (int64_t base = (((int64_t)blk * blockDim.x + threadIdx.x) - lane) * SCAN_V; base < a.n_rows; base += stride * SCAN_V) {
    const int64_t r0 = base + (int64_t)lane * SCAN_V;
    const uint32_t f = tiled_load_states<SCAN_V>(a.flags, r0, a.n_rows);
    if (!__ballot(tiled_has_state(f, 2u))) continue;
#pragma unroll
    for (int j = 0; j < SCAN_V; ++j) {
      const uint64_t mask = __ballot(((f >> (8 * j)) & 0xFFu) == 2u);
      if (mask) tiled_rows<0>(a, mask, (int)(r0 + j), 0u, step, alpha, 0.f, lane);
    }
  }
}
// rows whose state byte is 2: the step's ApplyAdam from their gradient.  One lane per row scans the state bytes.
template <int SCAN_V>
__global__ __launch_bounds__(256) void adam_touched_kernel(const TiledArgs a, uint32_t step, float alpha, int drop) {
  adam_touched_body<SCAN_V>(a, step, alpha, drop, (int)blockIdx.x, (int)gridDim.x);
}

// The step's whole ApplyAdam in ONE launch (round 4): blocks [0, nb_t) the touched rows of the table, blocks [nb_t, nb_t + nb_d)
// the flat dense variables (what score_adam does: the L2 term folded in for the regularised range) -- the two touch disjoint
// memory and were two dependent launches at the end of every step.  The guard is the table's (score_adam_table_t.id_status):
// set, neither half applies anything and the dense half counts the suppressed step.
struct DenseAdamArgs { float* p; float* m; float* v; const float* g; int64_t n4, n, n_reg; float l2; int32_t* skipped; };
template <int SCAN_V>
__global__ __launch_bounds__(256) void adam_step_kernel(const TiledArgs a, uint32_t step, float alpha, const DenseAdamArgs d,
                                                        int nb_t, int nb_d) {
  if ((int)blockIdx.x < nb_t) {
    adam_touched_body<SCAN_V>(a, step, alpha, 0, (int)blockIdx.x, nb_t);
    return;
  }
  if (tiled_guarded(a)) {
    if (d.skipped && (int)blockIdx.x == nb_t && threadIdx.x == 0) atomicAdd(d.skipped, 1);
    return;
  }
  score_adam_dense_body(d.p, d.m, d.v, d.g, d.n4, d.n, d.n_reg, d.l2, alpha, a.omb1, a.omb2, a.eps, (int)blockIdx.x - nb_t, nb_d);
}

// The same update driven by a LIST of rows (the unique rows of the batch, score_index_plan with dedup == 2) instead of a
// scan of the state bytes: a group of LPR lanes per list entry, four entries per group and trip with their four rows'
// streams requested together.  Entries whose row is not in state 2 (the dummy row 0, whose uses carry no gradient) are
// skipped, so the list may be a superset of the rows with a gradient.
__global__ __launch_bounds__(256) void adam_touched_rows_kernel(const TiledArgs a, const int32_t* __restrict__ rows,
                                                                const int32_t* __restrict__ n_rows_dev, uint32_t step,
                                                                float alpha) {
  const int n = *n_rows_dev;
  if (tiled_guarded(a)) {                   // (as adam_touched_kernel)
    const uint32_t applied = tiled_applied(a, step);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
      const int r = rows[i];
      if (a.flags[r] == 2) tiled_unmark_row(a, r, applied);
    }
    return;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) a.ring[step % SCORE_ADAM_RING] = alpha;
  const int gpb = blockDim.x / a.LPR;
  const int ch4 = ((int)threadIdx.x % a.LPR) * 4;
  const int64_t ngroups = (int64_t)gridDim.x * gpb;
  constexpr int U = 4;
  for (int64_t i0 = ((int64_t)blockIdx.x * gpb + threadIdx.x / a.LPR) * U; i0 < n; i0 += ngroups * U) {
    int row[U];
    bool on[U];
#pragma unroll
    for (int u = 0; u < U; ++u) row[u] = rows[i0 + u < n ? i0 + u : n - 1];
#pragma unroll
    for (int u = 0; u < U; ++u) on[u] = i0 + u < n && ch4 < a.D && a.flags[row[u]] == 2;
    float4 p[U], m[U], v[U], g[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t e = (int64_t)row[u] * a.D + (ch4 < a.D ? ch4 : 0);
      p[u] = ld4(a.p + e); m[u] = ld4(a.m + e); v[u] = ld4(a.v + e); g[u] = ld4(a.g + e);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!on[u]) continue;
      const int64_t e = (int64_t)row[u] * a.D + ch4;
      score_adam1(p[u].x, m[u].x, v[u].x, g[u].x, a.omb1, a.omb2, alpha, a.eps);
      score_adam1(p[u].y, m[u].y, v[u].y, g[u].y, a.omb1, a.omb2, alpha, a.eps);
      score_adam1(p[u].z, m[u].z, v[u].z, g[u].z, a.omb1, a.omb2, alpha, a.eps);
      score_adam1(p[u].w, m[u].w, v[u].w, g[u].w, a.omb1, a.omb2, alpha, a.eps);
      st4(a.p + e, p[u]); st4(a.m + e, m[u]); st4(a.v + e, v[u]);
      if (ch4 == 0) tiled_publish_applied(a, row[u], step);
    }
  }
}

// live rows of [row_begin, row_end) that lag behind `upto`: replay what they missed.  State 2 rows are left alone:
// they belong to the step in flight (score_adam_touched), and are current up to the step before by construction.
__global__ __launch_bounds__(256) void adam_catchup_rows_kernel(const TiledArgs a, int64_t row_begin, int64_t row_end,
                                                                uint32_t upto) {
  if (tiled_guarded(a)) return;
  const int lane = tiled_lane();
  const float areg = a.ring[(upto - (uint32_t)lane) % SCORE_ADAM_RING];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t base = row_begin + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane; base < row_end; base += stride) {
    const int64_t r = base + lane;
    uint32_t old = upto;
    bool hit = false;
    if (r < row_end && a.flags[r] == 1) {
      // (a plain load: a row of the batch in flight was brought up to `upto` = step - 1 before its forward pass, so even a stale
      //  count of such a row is not below upto -- past-the-cache loads made this kernel 20 -> 41 us)
      old = a.step[r];
      hit = old < upto;
    }
    const uint64_t mask = __ballot(hit);
    tiled_rows<1>(a, mask, (int)r, old, upto, 0.f, areg, lane);
  }
}

// score_adam_catchup_ids, first half: every value of ids[] that names a live row lagging behind `upto` puts that
// row into state 3.  Plain byte stores, no claim: a padding id or a hot categorical row occurs 10^5..10^6 times in a
// batch, and that many atomics on one address took 1.1 ms (an atomicMax claim on row_step was the first version);
// same-address stores of a wave merge, and waves that arrive after the first store has landed read 3 and skip theirs.
// Values outside [0, n_rows) are ignored, so the caller may pass a whole flat batch buffer (the lengths and labels
// in it name low rows: catching a row up early is always valid).
__global__ __launch_bounds__(256) void adam_mark_ids_kernel(const TiledArgs a, const int32_t* __restrict__ ids, int64_t n,
                                                            uint32_t upto, int set_alpha, float alpha_upto) {
  if (tiled_guarded(a)) return;
  // score_adam_catchup_ids_through: the replay that follows reads alpha of step `upto` from the ring before the step's own
  // score_adam_touched has written it (the same value)
  if (set_alpha && blockIdx.x == 0 && threadIdx.x == 0) a.ring[upto % SCORE_ADAM_RING] = alpha_upto;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int row = ids[i];
    // (the count twice: a plain load filters the occurrences of rows that are up to date -- a cached count is never AHEAD of the
    //  row's true one -- and only what passes pays the load past the caches; all 4.3 M ids of a cfg-3 batch that way was 18 -> 61 us)
    if (row >= 0 && (int64_t)row < a.n_rows && a.flags[row] == 1 && a.step[row] < upto && tiled_row_step(a, row) < upto)
      a.flags[row] = 3;
  }
}
// second half: the rows in state 3 are replayed up to `upto` and return to state 1 (the scan of score_adam_touched)
template <int SCAN_V>
__global__ __launch_bounds__(256) void adam_catchup_marked_kernel(const TiledArgs a, uint32_t upto) {
  if (tiled_guarded(a)) {                   // (the word was raised between the two halves: the marks go, nothing is replayed)
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_rows; r += (int64_t)gridDim.x * blockDim.x)
      if (a.flags[r] == 3) a.flags[r] = 1;
    return;
  }
  const int lane = tiled_lane();
  const float areg = a.ring[(upto - (uint32_t)lane) % SCORE_ADAM_RING];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t base = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane) * SCAN_V; base < a.n_rows; base += stride * SCAN_V) {
    const int64_t r0 = base + (int64_t)lane * SCAN_V;
    const uint32_t f = tiled_load_states<SCAN_V>(a.flags, r0, a.n_rows);
    if (!__ballot(tiled_has_state(f, 3u))) continue;
#pragma unroll
    for (int j = 0; j < SCAN_V; ++j) {
      const bool hit = ((f >> (8 * j)) & 0xFFu) == 3u;
      const uint64_t mask = __ballot(hit);
      if (!mask) continue;
      const uint32_t old = hit ? a.step[r0 + j] : upto;
      tiled_rows<2>(a, mask, (int)(r0 + j), old, upto, 0.f, areg, lane);
    }
  }
}

static int tiled_args(const score_adam_table_t* t, TiledArgs* a, bool need_g) {
  if (!t || !t->p || !t->m || !t->v || (need_g && !t->g) || !t->row_flags || !t->row_step || !t->alpha_ring ||
      t->n_rows <= 0 || t->n_rows > 0x7fffffffLL || t->D <= 0)
    return SCORE_E_BADARG;
  if ((t->D & 3) || t->D > 256) return SCORE_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(t->p) | reinterpret_cast<uintptr_t>(t->m) | reinterpret_cast<uintptr_t>(t->v) |
       reinterpret_cast<uintptr_t>(t->g)) & 15)
    return SCORE_E_SHAPE;
  a->p = t->p; a->m = t->m; a->v = t->v; a->g = t->g;
  a->flags = t->row_flags; a->step = t->row_step; a->ring = t->alpha_ring;
  a->n_rows = t->n_rows; a->D = t->D;
  int LPR = 1;
  while (LPR < t->D / 4) LPR <<= 1;
  a->LPR = LPR;
  a->omb1 = 1.0f - t->beta1; a->omb2 = 1.0f - t->beta2; a->eps = t->eps;
  a->guard = t->id_status;
  a->skipped = t->skipped_steps;
  return 0;
}
// State bytes per lane (SCAN_V): ONE -- except narrow rows in a table of more than 3 M rows, where a dword of four pays.  Measured
// inside the step, where the rows are cold (alone on warm rows, tools/adam_touched_probe.py, the two tie): with four a wave owns
// 256 rows, meets several hits and works through them one dependent memory round trip after the other -- Tmall default
// (1.5 M rows) 15.4 us (1) vs 27.4 us (4) for the step's ApplyAdam launch, cfg-3 (D = 64) 0.069 vs 0.153 ms -- but over the 5 M-row
// tables the 84 K single-byte waves of the two scans cost more than that: CCMR default 0.3540 (1) vs 0.3479 ms/step (4), Taobao
// default 0.2261 vs 0.2205 (two alternating pairs each, tools/build_variant.py variants).
#if defined(TILED_PROBE_SCAN4)          // timing probes: force one or the other
static int tiled_scan_v(const TiledArgs& a) { return a.D <= 32 ? 4 : 1; }
#elif defined(TILED_PROBE_SCAN1)
static int tiled_scan_v(const TiledArgs& a) { (void)a; return 1; }
#else
static int tiled_scan_v(const TiledArgs& a) { return (a.D <= 32 && a.n_rows > 3000000) ? 4 : 1; }
#endif
static int tiled_scan_blocks(const TiledArgs& a) {      // SCAN_V rows per thread
  const int64_t want = cdiv64(a.n_rows, 256 * tiled_scan_v(a));
  return (int)(want < 1 ? 1 : want < 32768 ? want : 32768);
}
static int tiled_blocks(int64_t n) {
  const int64_t want = cdiv64(n, 256);
  return (int)(want < 1 ? 1 : want < 32768 ? want : 32768);
}

extern "C" int score_adam_touched(const score_adam_table_t* t, uint32_t step, float alpha, void* stream) {
  TiledArgs a;
  SCORE_TRY(tiled_args(t, &a, true));
  if (step == 0) return SCORE_E_BADARG;
  if (tiled_scan_v(a) == 4) hipLaunchKernelGGL(adam_touched_kernel<4>, dim3(tiled_scan_blocks(a)), dim3(256), 0, (hipStream_t)stream, a, step, alpha, 0);
  else hipLaunchKernelGGL(adam_touched_kernel<1>, dim3(tiled_scan_blocks(a)), dim3(256), 0, (hipStream_t)stream, a, step, alpha, 0);
  SCORE_CHECK_LAUNCH();
  return 0;
}

extern "C" int score_adam_touched_and_dense(const score_adam_table_t* t, uint32_t step, float alpha, float* p, float* m, float* v,
                                            const float* g, int64_t n, int64_t n_reg, float l2, int32_t* skipped, void* stream) {
  TiledArgs a;
  SCORE_TRY(tiled_args(t, &a, true));
  if (step == 0 || !p || !m || !v || !g || n <= 0) return SCORE_E_BADARG;
  if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
       reinterpret_cast<uintptr_t>(g)) & 15)
    return SCORE_E_SHAPE;
  DenseAdamArgs d;
  d.p = p; d.m = m; d.v = v; d.g = g; d.n4 = n / 4; d.n = n; d.n_reg = n_reg; d.l2 = l2; d.skipped = skipped;
  const int nb_t = tiled_scan_blocks(a);
  const int64_t want = cdiv64(d.n4 > 0 ? d.n4 : 1, 256);
  const int nb_d = (int)(want < 8192 ? want : 8192);
  if (tiled_scan_v(a) == 4) hipLaunchKernelGGL(adam_step_kernel<4>, dim3(nb_t + nb_d), dim3(256), 0, (hipStream_t)stream, a, step, alpha, d, nb_t, nb_d);
  else hipLaunchKernelGGL(adam_step_kernel<1>, dim3(nb_t + nb_d), dim3(256), 0, (hipStream_t)stream, a, step, alpha, d, nb_t, nb_d);
  SCORE_CHECK_LAUNCH();
  return 0;
}

extern "C" int score_adam_unmark(const score_adam_table_t* t, uint32_t upto, void* stream) {
  TiledArgs a;
  SCORE_TRY(tiled_args(t, &a, false));
  hipLaunchKernelGGL(adam_touched_kernel<1>, dim3(tiled_blocks(a.n_rows)), dim3(256), 0, (hipStream_t)stream, a, upto + 1, 0.f, 1);
  SCORE_CHECK_LAUNCH();
  return 0;
}

extern "C" int score_adam_touched_rows(const score_adam_table_t* t, const int32_t* rows, const int32_t* n_rows_dev,
                                       int64_t max_rows, uint32_t step, float alpha, void* stream) {
  TiledArgs a;
  SCORE_TRY(tiled_args(t, &a, true));
  if (step == 0 || !rows || !n_rows_dev || max_rows <= 0) return SCORE_E_BADARG;
  const int gpb = 256 / a.LPR;
  int64_t blocks = cdiv64(cdiv64(max_rows, 4), gpb);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adam_touched_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, rows, n_rows_dev,
                     step, alpha);
  SCORE_CHECK_LAUNCH();
  return 0;
}

extern "C" int score_adam_catchup_rows(const score_adam_table_t* t, int64_t row_begin, int64_t row_end, uint32_t upto,
                                       void* stream) {
  TiledArgs a;
  SCORE_TRY(tiled_args(t, &a, false));
  if (row_begin < 0 || row_end > a.n_rows || row_begin > row_end) return SCORE_E_BADARG;
  if (row_begin == row_end) return 0;
  hipLaunchKernelGGL(adam_catchup_rows_kernel, dim3(tiled_blocks(row_end - row_begin)), dim3(256), 0, (hipStream_t)stream,
                     a, row_begin, row_end, upto);
  SCORE_CHECK_LAUNCH();
  return 0;
}

extern "C" int score_adam_catchup_ids(const score_adam_table_t* t, const int32_t* ids, int64_t n_ids, uint32_t upto,
                                      void* stream) {
  TiledArgs a;
  SCORE_TRY(tiled_args(t, &a, false));
  if (!ids || n_ids < 0) return SCORE_E_BADARG;
  if (n_ids == 0) return 0;
  if (upto == 0) return 0;         // nothing has been applied yet: nothing can lag
  hipLaunchKernelGGL(adam_mark_ids_kernel, dim3(tiled_blocks(cdiv64(n_ids, 4))), dim3(256), 0, (hipStream_t)stream, a, ids,
                     n_ids, upto, 0, 0.f);
  SCORE_CHECK_LAUNCH();
  if (tiled_scan_v(a) == 4) hipLaunchKernelGGL(adam_catchup_marked_kernel<4>, dim3(tiled_scan_blocks(a)), dim3(256), 0, (hipStream_t)stream, a, upto);
  else hipLaunchKernelGGL(adam_catchup_marked_kernel<1>, dim3(tiled_scan_blocks(a)), dim3(256), 0, (hipStream_t)stream, a, upto);
  SCORE_CHECK_LAUNCH();
  return 0;
}

extern "C" int score_adam_catchup_ids_through(const score_adam_table_t* t, const int32_t* ids, int64_t n_ids, uint32_t step,
                                              float alpha, void* stream) {
  TiledArgs a;
  SCORE_TRY(tiled_args(t, &a, false));
  if (!ids || n_ids < 0 || step == 0) return SCORE_E_BADARG;
  if (n_ids == 0) return 0;
  hipLaunchKernelGGL(adam_mark_ids_kernel, dim3(tiled_blocks(cdiv64(n_ids, 4))), dim3(256), 0, (hipStream_t)stream, a, ids,
                     n_ids, step, 1, alpha);
  SCORE_CHECK_LAUNCH();
  if (tiled_scan_v(a) == 4) hipLaunchKernelGGL(adam_catchup_marked_kernel<4>, dim3(tiled_scan_blocks(a)), dim3(256), 0, (hipStream_t)stream, a, step);
  else hipLaunchKernelGGL(adam_catchup_marked_kernel<1>, dim3(tiled_scan_blocks(a)), dim3(256), 0, (hipStream_t)stream, a, step);
  SCORE_CHECK_LAUNCH();
  return 0;
}
