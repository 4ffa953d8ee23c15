// The occurrence sort of the index plan, by hand, for SMALL batches (round 4): a stable LSD radix sort of (key, value) pairs
// of 32-bit words with 11- / 12-bit digits.
//
// Where it is used, and why only there.  For the reference's own batch sizes (B = 100 / 200: 0.15 - 1 M occurrences) rocPRIM
// sorts with NINETEEN kernels (block sort + a chain of merge passes), and the step is bound by the host's launch calls (~6 us
// each, ~57 per step): in the kernel trace the launch stream idles 140 us between the forward and the backward pass while
// the host queues the sort.  This sort is SIX launches whatever n (fill + histogram fused, column scan, scatter; histogram,
// column scan, scatter).  For cfg-3's 2.9 M occurrences the library switches to onesweep (three passes of 8 bits, 12
// launches) and the device side decides: there this sort is no faster alone (139 vs ~145 us) and 1.5 % SLOWER inside the
// step -- 2,048 bins leave four-pair runs per tile and bin, so its stores are 16-byte pieces, and its 78-KB-LDS workgroups
// crowd the gather and the recurrence beside it (profiles/r04_probes.md); at cfg-5's 23.6 / 57 M it ties (965 / 2,170 us).
// So score_launch_plan (scatter.hip) takes it below SCORE_OWN_SORT_MAX_N occurrences and the library above; both are stable,
// so the plan -- and every sum the pull scatter builds from it -- is bit for bit the same either way
// (score_state_t.debug_flags bits 5 / 8 force one or the other: tests/test_gpu_ops.py, test_gpu_model.py).
//
// 160 KB of LDS per CU hold a 2,048- or 4,096-bin ranking of an 8,192-pair tile, so a 21-bit key needs TWO passes of 11 bits
// (23 bits: 12 + 11), and the first pass's histogram comes out of the kernel that writes the keys in the first place.
//
// One pass = three kernels.
//   hist     M[tile][bin] = how many of the tile's 8,192 keys have that digit (LDS counters; for pass 1 fused into the fill)
//   colscan  per bin: exclusive prefix over the tiles, in place, and the bin's total
//   scatter  per tile: stable rank of every key among the tile's keys of the same digit -- eight waves, each ranking its
//            1,024 consecutive keys in 16 rounds of 64 by wave-wide digit matching (dbits ballots: deterministic, no LDS
//            atomics, so equal keys keep their order) into per-(digit, wave) counters --, then the tile is put in digit order
//            in LDS and written out in that order: a run of equal digits goes to consecutive addresses.
// Destination of the pair at tile-local sorted position i with digit d:
//   binbase[d] (keys with a smaller digit, all tiles) + M[tile][d] (same digit, earlier tiles) + i - dpre[d] (same digit,
//   this tile, before it).
#include <atomic>
#include "common.h"
#include "kernels.h"

namespace {

constexpr int TILE = 8192, THREADS = 512, WAVES = 8, EPT = 16, PER_WAVE = TILE / WAVES;   // EPT rounds of 64 keys per wave

// exclusive prefix of x over the 512 threads of the block (all of them call); sc: 16 words of LDS
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t x, uint32_t* sc, int tid) {
  const int lane = tid & 63, w = tid >> 6;
  uint32_t incl = x;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = (uint32_t)__shfl_up((int)incl, off, SCORE_WAVE);
    if (lane >= off) incl += y;
  }
  __syncthreads();                       // (sc may still be read by a previous call)
  if (lane == 63) sc[w] = incl;
  __syncthreads();
  uint32_t woff = 0;
#pragma unroll
  for (int i = 0; i < WAVES; ++i) woff += i < w ? sc[i] : 0u;
  return woff + incl - x;
}

// hist[d] += 1 for the lanes with `ok`, without the same-address pile-up: the digits of a wave's 64 keys are anything but
// uniform (a third are the dummy row, most of the rest a dozen hot categorical rows -- and in the second pass all of those
// share a handful of high digits), and 64 LDS atomics on one address serialise (the first version: 62 us for an 11.5 MB
// read).  Up to three rounds take the digit of the first lane still active, count its holders with one ballot and add the
// count once; whoever is left adds for itself.
__device__ __forceinline__ void hist_add(uint32_t* hist, uint32_t d, bool ok, int lane) {
  uint64_t act = __ballot(ok);
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    if (!act) break;                                               // (wave-uniform)
    const int first = __ffsll((unsigned long long)act) - 1;
    const uint32_t d0 = (uint32_t)__shfl((int)d, first, SCORE_WAVE);
    const uint64_t m = __ballot(ok && d == d0) & act;
    if (lane == first) atomicAdd(&hist[d0], (uint32_t)__popcll(m));
    act &= ~m;
  }
  if ((act >> lane) & 1ull) atomicAdd(&hist[d], 1u);
}

// ---------------------------------------------------------------- plan fill (score_index_plan's occurrence enumeration)
// occurrence descriptor: seg[31:29] f[28:26] k[25:21] bt[20:0]
#define DESC(seg, f, k, bt) (((uint32_t)(seg) << 29) | ((uint32_t)(f) << 26) | ((uint32_t)(k) << 21) | (uint32_t)(bt))

__device__ __forceinline__ void plan_fill_one(const PlanFillArgs& a, int64_t i, uint32_t& key, uint32_t& val) {
  if (i == a.off[6]) {  // sentinel occurrence of the dummy row: unique position 0 is always row 0
    key = 0;
    val = DESC(7, 0, 0, 0);
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
  key = row;
  if (a.G > 1) key = ((row % a.G) << a.shift) | (row / a.G);   // (owner, local row)
  val = DESC(seg, f, k, bt);
}

// keys / vals of the tile written, and the tile's histogram of the FIRST pass's digit
__device__ __forceinline__ void plan_fill_hist_tile(const PlanFillArgs& a, int64_t n, uint32_t* __restrict__ keys,
                                                    uint32_t* __restrict__ vals, int dbits, uint32_t* __restrict__ M,
                                                    int tile, uint32_t* hist) {
  const int nbins = 1 << dbits, tid = threadIdx.x;
  for (int d = tid; d < nbins; d += THREADS) hist[d] = 0;
  __syncthreads();
  const int64_t base = (int64_t)tile * TILE;
  const int lane = tid & 63;
#pragma unroll 4
  for (int j = 0; j < EPT; ++j) {
    const int64_t i = base + j * THREADS + tid;
    uint32_t key = 0, val = 0;
    if (i < n) {
      plan_fill_one(a, i, key, val);
      keys[i] = key;
      vals[i] = val;
    }
    hist_add(hist, key & (uint32_t)(nbins - 1), i < n, lane);
  }
  __syncthreads();
  uint32_t* row = M + (int64_t)tile * nbins;
  for (int d = tid; d < nbins; d += THREADS) row[d] = hist[d];
}
__global__ __launch_bounds__(THREADS) void plan_fill_hist_kernel(PlanFillArgs a, int64_t n, uint32_t* __restrict__ keys,
                                                                 uint32_t* __restrict__ vals, int dbits,
                                                                 uint32_t* __restrict__ M) {
  extern __shared__ uint32_t hist[];
  plan_fill_hist_tile(a, n, keys, vals, dbits, M, (int)blockIdx.x, hist);
}

__device__ __forceinline__ void sort_hist_tile(const uint32_t* __restrict__ keys, int64_t n, int shift, int dbits,
                                               uint32_t* __restrict__ M, int tile, uint32_t* hist) {
  const int nbins = 1 << dbits, tid = threadIdx.x;
  for (int d = tid; d < nbins; d += THREADS) hist[d] = 0;
  __syncthreads();
  const int64_t base = (int64_t)tile * TILE;
  uint32_t k[EPT];
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    const int64_t i = base + j * THREADS + tid;
    k[j] = i < n ? keys[i] : 0u;
  }
#pragma unroll
  for (int j = 0; j < EPT; ++j)
    hist_add(hist, (k[j] >> shift) & (uint32_t)(nbins - 1), base + j * THREADS + tid < n, tid & 63);
  __syncthreads();
  uint32_t* row = M + (int64_t)tile * nbins;
  for (int d = tid; d < nbins; d += THREADS) row[d] = hist[d];
}
__global__ __launch_bounds__(THREADS) void sort_hist_kernel(const uint32_t* __restrict__ keys, int64_t n, int shift, int dbits,
                                                            uint32_t* __restrict__ M) {
  extern __shared__ uint32_t hist[];
  sort_hist_tile(keys, n, shift, dbits, M, (int)blockIdx.x, hist);
}

// per bin: M[tile][bin] <- number of keys with that digit in EARLIER tiles; tot[bin] <- in all tiles.
// A block of 16 waves takes 64 bins (a lane per bin: 256-byte rows of M), a wave a contiguous range of tiles.
constexpr int CS_WAVES = 16;
__global__ __launch_bounds__(CS_WAVES * 64) void sort_colscan_kernel(uint32_t* __restrict__ M, int ntiles, int nbins,
                                                                    uint32_t* __restrict__ tot) {
  __shared__ uint32_t wsum[CS_WAVES][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int bin = blockIdx.x * 64 + lane;
  const bool on = bin < nbins;
  const int chunk = (ntiles + CS_WAVES - 1) / CS_WAVES;
  const int t0 = w * chunk, t1 = min(ntiles, t0 + chunk);
  uint32_t* col = M + (on ? bin : 0);
  uint32_t s = 0;
  int t = t0;
  for (; t + 8 <= t1; t += 8) {
    uint32_t c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = col[(int64_t)(t + u) * nbins];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += c[u];
  }
  for (; t < t1; ++t) s += col[(int64_t)t * nbins];
  wsum[w][lane] = s;
  __syncthreads();
  uint32_t run = 0, total = 0;
#pragma unroll
  for (int i = 0; i < CS_WAVES; ++i) {
    const uint32_t x = wsum[i][lane];
    run += i < w ? x : 0u;
    total += x;
  }
  if (w == 0 && on) tot[bin] = total;
  if (!on) return;
  t = t0;
  for (; t + 8 <= t1; t += 8) {
    uint32_t c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = col[(int64_t)(t + u) * nbins];
#pragma unroll
    for (int u = 0; u < 8; ++u) { col[(int64_t)(t + u) * nbins] = run; run += c[u]; }
  }
  for (; t < t1; ++t) { const uint32_t c = col[(int64_t)t * nbins]; col[(int64_t)t * nbins] = run; run += c; }
}

// LDS of the scatter kernel, in 32-bit words: cnt u16[nbins][8] | dpre u16[nbins] | adj i32[nbins] | ex u32[TILE] | sc[16]
__host__ __device__ constexpr int scatter_lds_bytes(int nbins) { return nbins * 16 + nbins * 2 + nbins * 4 + TILE * 4 + 64; }

__device__ __forceinline__ void sort_scatter_tile(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                  uint32_t* __restrict__ kout, uint32_t* __restrict__ vout, int64_t n, int shift,
                                                  int dbits, const uint32_t* __restrict__ M, const uint32_t* __restrict__ tot,
                                                  int tile, uint32_t* smem, int raw_tiles = 0) {
  const int nbins = 1 << dbits;
  const uint32_t mask = (uint32_t)(nbins - 1);
  uint16_t* cnt = reinterpret_cast<uint16_t*>(smem);           // [d][wave]: count, later the exclusive prefix over waves
  uint16_t* dpre = cnt + nbins * WAVES;                         // tile-local exclusive prefix over digits
  int32_t* adj = reinterpret_cast<int32_t*>(dpre + nbins);      // destination of sorted position i with digit d: adj[d] + i
  uint32_t* ex = reinterpret_cast<uint32_t*>(adj + nbins);
  uint32_t* sc = ex + TILE;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t base = (int64_t)tile * TILE;
  const int ntile = (int)min((int64_t)TILE, n - base);
  __syncthreads();                       // (the fused kernel: a previous phase of this workgroup may still read the LDS)
  for (int i = tid; i < nbins * WAVES / 2; i += THREADS) smem[i] = 0;
  uint32_t key[EPT], val[EPT], pos[EPT];
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int e = w * PER_WAVE + r * 64 + lane;
    const bool ok = e < ntile;
    key[r] = ok ? kin[base + e] : 0xFFFFFFFFu;
    val[r] = ok ? vin[base + e] : 0u;
  }
  __syncthreads();
  // stable rank inside (digit, wave): the lanes of a round that hold the same digit find each other by dbits ballots
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const bool ok = w * PER_WAVE + r * 64 + lane < ntile;
    const uint32_t d = (key[r] >> shift) & mask;
    uint64_t peers = __ballot(ok);
    for (int b = 0; b < dbits; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t m = __ballot(ok && bit);
      peers &= bit ? m : ~m;
    }
    if (!ok) peers = 1ull << lane;                         // (a lane past the end: alone, and it writes nothing)
    const int leader = __ffsll((unsigned long long)peers) - 1;
    const uint32_t below = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
    uint32_t old = 0;
    if (ok && lane == leader) {
      old = cnt[d * WAVES + w];
      cnt[d * WAVES + w] = (uint16_t)(old + (uint32_t)__popcll(peers));
    }
    old = (uint32_t)__shfl((int)old, leader, SCORE_WAVE);
    pos[r] = old + below;
  }
  __syncthreads();
  // per digit: exclusive prefix over the waves (in place) and the tile's total; then the prefix over digits, of this tile
  // (dpre) and of all tiles (binbase, from the column scan's totals)
  const int dpt = nbins >= THREADS ? nbins / THREADS : 1;          // digits per thread, contiguous
  const bool has = tid * dpt < nbins;
  // raw_tiles > 0 (small sorts, round 5): M holds the tiles' raw histograms and this workgroup takes its own column prefix
  // (same digit, earlier tiles) and the bins' totals from them -- raw_tiles x dpt words per thread -- instead of a column-scan
  // launch in front of every scatter (two of the sort's six launches: the step is bound by the host's launch calls there)
  uint32_t lt[8], gt[8], mb[8], ls = 0, gs = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    lt[q] = gt[q] = mb[q] = 0;
    if (q < dpt && has) {
      const int d = tid * dpt + q;
      uint32_t run = 0;
#pragma unroll
      for (int i = 0; i < WAVES; ++i) { const uint32_t c = cnt[d * WAVES + i]; cnt[d * WAVES + i] = (uint16_t)run; run += c; }
      lt[q] = run;
      if (raw_tiles <= 0) {
        gt[q] = tot[d];
        mb[q] = M[(int64_t)tile * nbins + d];
      }
      ls += run;
    }
  }
  if (raw_tiles > 0 && has) {
    // four tiles' counts of this thread's digits in flight per trip (a load at a time made the scatter 25 -> 44 us)
    const uint32_t* col = M + tid * dpt;
    for (int t0 = 0; t0 < raw_tiles; t0 += 4) {
      uint32_t c[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t* row = col + (int64_t)(t0 + u < raw_tiles ? t0 + u : 0) * nbins;
#pragma unroll
        for (int q = 0; q < 8; ++q) c[u][q] = q < dpt ? row[q] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool in = t0 + u < raw_tiles, early = t0 + u < tile;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          gt[q] += in ? c[u][q] : 0u;
          mb[q] += early ? c[u][q] : 0u;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) gs += (q < dpt && has) ? gt[q] : 0u;
  uint32_t lo = block_excl_scan(ls, sc, tid);
  uint32_t go = block_excl_scan(gs, sc, tid);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    if (q < dpt && has) {
      const int d = tid * dpt + q;
      dpre[d] = (uint16_t)lo;
      adj[d] = (int32_t)(go + mb[q]) - (int32_t)lo;
      lo += lt[q];
      go += gt[q];
    }
  }
  __syncthreads();
  // the tile in digit order: keys through LDS, written out in sorted order (coalesced along runs of equal digits); then the values
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const bool ok = w * PER_WAVE + r * 64 + lane < ntile;
    if (ok) {
      const uint32_t d = (key[r] >> shift) & mask;
      pos[r] = (uint32_t)dpre[d] + (uint32_t)cnt[d * WAVES + w] + pos[r];
      ex[pos[r]] = key[r];
    }
  }
  __syncthreads();
  uint32_t dest[EPT];
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    const int i = j * THREADS + tid;
    dest[j] = 0;
    if (i < ntile) {
      const uint32_t k = ex[i];
      dest[j] = (uint32_t)(adj[(k >> shift) & mask] + i);
      kout[dest[j]] = k;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < EPT; ++r)
    if (w * PER_WAVE + r * 64 + lane < ntile) ex[pos[r]] = val[r];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    const int i = j * THREADS + tid;
    if (i < ntile) vout[dest[j]] = ex[i];
  }
}
__global__ __launch_bounds__(THREADS) void sort_scatter_kernel(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                               uint32_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                               int64_t n, int shift, int dbits, const uint32_t* __restrict__ M,
                                                               const uint32_t* __restrict__ tot, int raw_tiles) {
  extern __shared__ uint32_t smem[];
  sort_scatter_tile(kin, vin, kout, vout, n, shift, dbits, M, tot, (int)blockIdx.x, smem, raw_tiles);
}

// (up to FUSED_MAX_TILES tiles the scatter kernel scans its own histogram columns: four launches instead of six)
constexpr int FUSED_MAX_TILES = 64;
struct SortShape { int npass, dbits[3], shift[3]; };
SortShape sort_shape(int key_bits) {
  SortShape s;
  if (key_bits < 1) key_bits = 1;
  if (key_bits > 32) key_bits = 32;
  s.npass = (key_bits + 11) / 12;                       // 12 bits per pass at most (4,096 bins)
  int left = key_bits, sh = 0;
  for (int p = 0; p < s.npass; ++p) {
    const int b = (left + (s.npass - p) - 1) / (s.npass - p);
    s.dbits[p] = b; s.shift[p] = sh;
    sh += b; left -= b;
  }
  return s;
}

// more than 64 KB of dynamic LDS needs the function attribute, once per device (the largest request so far is remembered)
int set_lds(const void* fn, int bytes) {
  if (bytes <= 64 * 1024) return 0;
  static std::atomic<int> have[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SCORE_E_BADARG;
  if (have[dev].load(std::memory_order_acquire) >= bytes) return 0;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  have[dev].store(bytes, std::memory_order_release);
  return 0;
}

int colscan_and_scatter(const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, int64_t n, int shift, int dbits,
                        uint32_t* M, uint32_t* tot, hipStream_t s) {
  const int nbins = 1 << dbits;
  const int ntiles = (int)cdiv64(n, TILE);
  const int raw = ntiles <= FUSED_MAX_TILES ? ntiles : 0;       // (up to 64 tiles the scatter scans its own columns: four launches)
  if (!raw) {
    hipLaunchKernelGGL(sort_colscan_kernel, dim3((nbins + 63) / 64), dim3(CS_WAVES * 64), 0, s, M, ntiles, nbins, tot);
    SCORE_CHECK_LAUNCH();
  }
  const int lds = scatter_lds_bytes(nbins);
  SCORE_TRY(set_lds((const void*)sort_scatter_kernel, lds));
  hipLaunchKernelGGL(sort_scatter_kernel, dim3(ntiles), dim3(THREADS), lds, s, kin, vin, kout, vout, n, shift, dbits, M, tot, raw);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// passes first .. npass-1 of the sort; `hist_done`: M already holds pass `first`'s histogram.  Ping-pongs between (k0, v0) and
// (k1, v1) starting from (k0, v0); the result is in (k0, v0) after an even number of passes, in (k1, v1) after an odd one.
int run_passes(const SortShape& sh, uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, int64_t n, uint32_t* M, uint32_t* tot,
               bool hist_done, hipStream_t s) {
  const int ntiles = (int)cdiv64(n, TILE);
  for (int p = 0; p < sh.npass; ++p) {
    const int nbins = 1 << sh.dbits[p];
    if (!(p == 0 && hist_done)) {
      hipLaunchKernelGGL(sort_hist_kernel, dim3(ntiles), dim3(THREADS), nbins * 4, s, k0, n, sh.shift[p], sh.dbits[p], M);
      SCORE_CHECK_LAUNCH();
    }
    SCORE_TRY(colscan_and_scatter(k0, v0, k1, v1, n, sh.shift[p], sh.dbits[p], M, tot, s));
    uint32_t* t = k0; k0 = k1; k1 = t;
    t = v0; v0 = v1; v1 = t;
  }
  return 0;
}

}  // namespace

size_t score_sort_temp_bytes(int64_t n) {
  // sized for 4,096 bins whatever the key width: the histogram matrix [tiles][bins] and the bins' totals
  const int64_t ntiles = cdiv64(n > 0 ? n : 1, TILE);
  return (size_t)((ntiles + 1) * 4096 * 4);
}

// keys_out / vals_out <- the occurrences of the batch sorted by (owner, row), equal keys in occurrence order (stable).
// keys_in / vals_in are scratch (they hold an intermediate pass afterwards).
int score_launch_plan_own(const PlanFillArgs& a, int key_bits, uint32_t* keys_in, uint32_t* vals_in, uint32_t* keys_out,
                      uint32_t* vals_out, void* temp, size_t temp_bytes, hipStream_t s) {
  const int64_t n = a.off[6] + 1;   // + sentinel
  if (n >= (1ll << 31)) return SCORE_E_SHAPE;
  if (score_sort_temp_bytes(n) > temp_bytes) return SCORE_E_WORKSPACE;
  const SortShape sh = sort_shape(key_bits);
  const int ntiles = (int)cdiv64(n, TILE);
  uint32_t* M = reinterpret_cast<uint32_t*>(temp);
  uint32_t* tot = M + (int64_t)ntiles * 4096;
  // an even number of passes ends where it started: the fill writes into the OUT arrays then
  uint32_t* k0 = (sh.npass & 1) ? keys_in : keys_out;
  uint32_t* v0 = (sh.npass & 1) ? vals_in : vals_out;
  uint32_t* k1 = (sh.npass & 1) ? keys_out : keys_in;
  uint32_t* v1 = (sh.npass & 1) ? vals_out : vals_in;
  hipLaunchKernelGGL(plan_fill_hist_kernel, dim3(ntiles), dim3(THREADS), (1 << sh.dbits[0]) * 4, s, a, n, k0, v0, sh.dbits[0], M);
  SCORE_CHECK_LAUNCH();
  return run_passes(sh, k0, v0, k1, v1, n, M, tot, true, s);
}

// C-ABI: the sort alone (include/score_hip.h)
extern "C" int64_t score_sort_pairs_temp_bytes(int64_t n) { return n > 0 ? (int64_t)score_sort_temp_bytes(n) : -1; }

extern "C" int score_sort_pairs(uint32_t* keys, uint32_t* vals, uint32_t* keys_alt, uint32_t* vals_alt, int64_t n,
                                int32_t key_bits, void* temp, int64_t temp_bytes, int32_t* result_in_alt, void* stream) {
  if (!keys || !vals || !keys_alt || !vals_alt || !temp || n <= 0 || key_bits < 1 || key_bits > 32) return SCORE_E_BADARG;
  if (n >= (1ll << 31)) return SCORE_E_SHAPE;
  if ((int64_t)score_sort_temp_bytes(n) > temp_bytes) return SCORE_E_WORKSPACE;
  const SortShape sh = sort_shape(key_bits);
  uint32_t* M = reinterpret_cast<uint32_t*>(temp);
  uint32_t* tot = M + cdiv64(n, TILE) * 4096;
  if (result_in_alt) *result_in_alt = sh.npass & 1;
  return run_passes(sh, keys, vals, keys_alt, vals_alt, n, M, tot, false, (hipStream_t)stream);
}
