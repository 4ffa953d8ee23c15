// gemm_panel.hip -- fp32-accurate bf16x3 products C[M,N] = A[M,K] . Bt[N,K]^T (+ bias) with a WHOLE-N output panel per
// workgroup: the GRU input projections x . [Wx_gates | Wx_cand] (score.py:205-208 through TF's GRUCell) and their input
// gradients d x = [dgates | dcand] . [Wx_gates | Wx_cand]^T, the two largest products of a step.
//
// gemm_bf16x3.hip tiles the output 128 x 128: every A panel is delivered into three CUs and every CU pulls 16 MAC per
// operand byte, and `tools/x3_probe.py` / `tools/x3p_probe.py` (profiles/r03_probes.md) show those products waiting on
// operand delivery, not on the matrix pipe.  Here a workgroup owns 16*MT rows x all N columns:
//   * A (activations, the big operand) is read from memory exactly once: fp32 rows -> registers (one k-tile ahead) ->
//     split into three bf16 planes (x = x0 + x1 + x2 exactly, as gemm_bf16x3.hip) -> LDS, two stages of 3 x 16*MT x 64 B with
//     the 16-B chunks XOR-swizzled, one barrier per 32-deep k-tile.  Every wave reads every A fragment.
//   * Bt (weights) never passes through LDS: score_gemm_panel_prep writes, once per step, its image
//     [k-tile][n-block][plane][lane] of 16-B MFMA fragments (lane l: column l % 16, k = 8 * (l / 16) .. +8); a wave owns
//     16 * NBW columns and loads its fragments with one coalesced 1-KB global_load_dwordx4 each, straight into registers,
//     re-loading a group of n-blocks for the next k-tile right behind its last use (so they have half a k-tile to land).
//   * v_mfma_f32_16x16x32_bf16 with the weight fragment as FIRST operand: the accumulator holds C^T, a lane has four
//     consecutive columns of one row -> float4 stores.  Six products per fragment pair (the terms above 2^-24), small first.
//   * no branch in the k loop and none per store: a load under a branch makes the compiler wait for vmcnt(0), i.e. for
//     the tile it has just requested.  Past the last k-tile the last one is loaded again (into the stage nobody reads);
//     rows past M are loaded as copies of row M - 1, hold its values and are stored onto it.
// 144 x 384 panel: 258 KB of A + 1.03 MB of weight fragments (L2 hits) for 24.8 M MAC = 19 MAC per delivered byte, and the
// 256 panels of cfg-3's two sides are one round of the chip.  Measured (tools/x3n_probe.py, both sides, 18,432 rows each):
// projection 68 us (101 with gemm_bf16x3.hip), input gradient 71 us (101).
#include "kernels.h"
#include <atomic>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4 gfloat4;      // (an integer cast back to a plain pointer would be a flat access)
typedef __attribute__((address_space(1))) u32x4 guint4;

// exact 3-way split (truncation): the three bf16 bit patterns sit in the upper halves of h, m, l
__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  const uint32_t xb = __float_as_uint(x);
  h = xb & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xFFFF0000u;
  l = __float_as_uint(r1 - __uint_as_float(m));
}
__device__ __forceinline__ uint32_t pack2(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
__device__ __forceinline__ uint64_t uni64(const void* p_) {      // a uniform pointer pinned in scalar registers
  const uint64_t p = reinterpret_cast<uint64_t>(p_);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
  return ((uint64_t)hi << 32) | lo;
}
// 64-B LDS rows of four 16-B chunks: chunk ^= {0,3,2,1}[(row >> 2) & 3] -- conflict-free for the 16-lane groups of
// ds_read_b128 ({0-3,12-15,20-27}, ...) on the 16x16x32 fragment (lane l: row l % 16, chunk l / 16) and for ds_write_b64
__device__ __forceinline__ int swz(int row) { return (0x6C >> (((row >> 2) & 3) * 2)) & 3; }

#if defined(PANEL_PROBE_NOMFMA)      // tools/x3n_probe.py builds this file with one ingredient stripped (wrong results; timing)
#define P_MFMA(a, b, c) (c)
#else
#define P_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif

struct PrepArgs { const float* B[4]; uint4* img[4]; int ldb, trans, N, K, NB; };

// img[((t * NB + j) * 3 + p) * 64 + lane], NB n-blocks incl. zero padding; Bt(n, k) = trans ? B[k * ldb + n] : B[n * ldb + k]
__global__ void panel_prep_kernel(const PrepArgs a) {
  const float* __restrict__ B = a.B[blockIdx.y];
  uint4* __restrict__ img = a.img[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per (t, j, lane)
  const int lane = (int)(i & 63);
  const int64_t tj = i >> 6;
  const int j = (int)(tj % a.NB), t = (int)(tj / a.NB);
  if (t * 32 >= a.K) return;
  const int n = j * 16 + (lane & 15), k0 = t * 32 + 8 * (lane >> 4);
  uint32_t p[3][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + e;
    const float x = n < a.N ? (a.trans ? B[(int64_t)k * a.ldb + n] : B[(int64_t)n * a.ldb + k]) : 0.f;
    split3(x, p[0][e], p[1][e], p[2][e]);
  }
#pragma unroll
  for (int q = 0; q < 3; ++q)
    img[(tj * 3 + q) * 64 + lane] = make_uint4(pack2(p[q][0], p[q][1]), pack2(p[q][2], p[q][3]), pack2(p[q][4], p[q][5]), pack2(p[q][6], p[q][7]));
}

struct PanelArgs { PanelGroup g[4]; int M, N, K, lda, ldc, tiles, NB; };

template <int MT, int NBW, bool BIAS>
__global__ __launch_bounds__(512, 1) void gemm_panel_kernel(const PanelArgs a) {
  constexpr int R = MT * 16, PLANE = R * 64, STAGE = 3 * PLANE;
  constexpr int NCH = R * 8;                          // float4 chunks of a k-tile of A
  constexpr int NRD = (NCH + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = blockIdx.x / a.tiles, tile = blockIdx.x - grp * a.tiles;
  const PanelGroup& G = a.g[grp];
  const int bm = tile * R;
  const int nt = a.K >> 5;
  const int lc = lane & 15, lq = lane >> 4;

  // A staging: chunk idx = tid + i * 512 -> row (tid >> 3) + 64 * i, float4 j = tid & 7.  Uniform base + 32-bit lane offsets
  // (the swizzle of a row depends on (row >> 2) & 3 only: rows 64 apart, and the rows of all m-blocks at one lane, share it)
  const int arow = tid >> 3, aj = tid & 7;
  const int rows_here = min(R, a.M - bm);              // (>= 1)
  uint32_t asrc[NRD];
#pragma unroll
  for (int i = 0; i < NRD; ++i) asrc[i] = (uint32_t)((min(arow + 64 * i, rows_here - 1) * a.lda + aj * 4) * 4);
  const uint32_t adst = (uint32_t)(arow * 64 + (((aj >> 1) ^ swz(arow)) * 16) + (aj & 1) * 8);
  const uint64_t abase = uni64(G.A + (int64_t)bm * a.lda);
  f32x4 areg[NRD];
  auto a_load = [&](int t) {
    const uint64_t b = abase + (uint64_t)t * 128;
#pragma unroll
    for (int i = 0; i < NRD; ++i)
#ifndef PANEL_PROBE_NOALOAD
      areg[i] = *(const gfloat4*)(b + asrc[i]);
#else
      areg[i] = f32x4{(float)t, 1.f, 2.f, (float)asrc[i]};
#endif
  };
  auto a_store = [&](int stage) {
    unsigned char* st = lds + stage * STAGE + adst;
#ifdef PANEL_PROBE_NOASTORE
    if (areg[0][0] != 123.456f) return;
#endif
#pragma unroll
    for (int i = 0; i < NRD; ++i) {
      uint32_t h[4], m[4], l[4];
      split3(areg[i][0], h[0], m[0], l[0]); split3(areg[i][1], h[1], m[1], l[1]);
      split3(areg[i][2], h[2], m[2], l[2]); split3(areg[i][3], h[3], m[3], l[3]);
      // (the last round covers the tile only in part: its other lanes write a dummy slot behind the stages)
      unsigned char* d = (i < NRD - 1 || tid + i * 512 < NCH) ? st + i * 4096 : lds + 2 * STAGE + tid * 8;
      *reinterpret_cast<uint2*>(d) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
      *reinterpret_cast<uint2*>(d + PLANE) = make_uint2(pack2(m[0], m[1]), pack2(m[2], m[3]));
      *reinterpret_cast<uint2*>(d + 2 * PLANE) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
    }
  };

  // this wave's weight fragments: uniform base per (k-tile, n-block, plane) + lane * 16.  (A wave past the last column
  // multiplies the image's zero padding: the same instruction stream for every wave)
  const int jb0 = wave * NBW;
  const uint64_t bbase = uni64(G.img) + (uint64_t)jb0 * 3 * 1024;
  const uint64_t bstep = (uint64_t)a.NB * 3 * 1024;       // bytes per k-tile
  const uint32_t boff = (uint32_t)lane * 16;
  bf16x8 bf[NBW][3];
  auto b_load = [&](int n, int t) {
    const uint64_t b = bbase + (uint64_t)t * bstep + (uint64_t)(n * 3) * 1024;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#ifndef PANEL_PROBE_NOBLOAD
      bf[n][p] = __builtin_bit_cast(bf16x8, *(const guint4*)(b + p * 1024 + boff));
#else
      bf[n][p] = __builtin_bit_cast(bf16x8, u32x4{(uint32_t)b, boff, (uint32_t)p, 0x3f803f80u});
#endif
  };

  f32x4 acc[NBW][MT];
#pragma unroll
  for (int n = 0; n < NBW; ++n)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment reads: row m * 16 + lc, logical chunk lq (one lane offset; m-block and plane are immediates)
  const uint32_t aoff = (uint32_t)(lc * 64 + ((lq ^ swz(lc)) * 16));

  // one k-tile of one group of n-blocks [n0, n1): A fragments read once per m-block and used for every n-block of the group
  auto mma_group = [&](int t, auto n0c, auto n1c) {
    constexpr int n0 = decltype(n0c)::value, n1 = decltype(n1c)::value;
    // (every group re-reads the A fragments: hidden from common-subexpression elimination, which would keep all 3 * MT
    //  of them alive across the groups and spill)
    uint32_t so = (uint32_t)(t & 1) * STAGE + aoff;
    asm volatile("" : "+v"(so));
    const unsigned char* sn = lds + so;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      bf16x8 af[3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#ifndef PANEL_PROBE_NOREAD
        af[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sn + p * PLANE + m * 1024));
#else
        af[p] = __builtin_bit_cast(bf16x8, u32x4{so, (uint32_t)(p + m), 0x3f803f80u, 0x3f803f80u});
#endif
#pragma unroll
      for (int n = n0; n < n1; ++n) {
        f32x4 c = acc[n][m];
        c = P_MFMA(bf[n][2], af[0], c);
        c = P_MFMA(bf[n][0], af[2], c);
        c = P_MFMA(bf[n][1], af[1], c);
        c = P_MFMA(bf[n][1], af[0], c);
        c = P_MFMA(bf[n][0], af[1], c);
        c = P_MFMA(bf[n][0], af[0], c);
        acc[n][m] = c;
      }
    }
  };
  constexpr int NG0 = NBW / 2;       // n-blocks of the first group
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, NG0> I1;
  typedef std::integral_constant<int, NBW> I2;

  // (the prologue issues its loads in the loop's order -- A tile, then the weight fragments -- so that the wait counts the
  //  compiler derives at the loop head are the steady state's, not a conservative merge)
  a_load(0);
  a_store(0);
  {
    a_load(min(1, nt - 1));
#pragma unroll
    for (int n = 0; n < NBW; ++n) b_load(n, 0);
    for (int t = 0; t < nt; ++t) {
      const int t1 = min(t + 1, nt - 1), t2 = min(t + 2, nt - 1);
      __syncthreads();                 // stage t & 1 is complete; nobody reads stage (t + 1) & 1 any more
      a_store((t + 1) & 1);
      a_load(t2);
      mma_group(t, I0(), I1());
#pragma unroll
      for (int n = 0; n < NG0; ++n) b_load(n, t1);        // behind their last use of this k-tile
      mma_group(t, I1(), I2());
#pragma unroll
      for (int n = NG0; n < NBW; ++n) b_load(n, t1);
    }
  }

  // C^T in the accumulators: lane = row lc of the m-block, columns 4*lq .. +4 of the n-block.  Stored from there, a wave
  // instruction writes 16 rows x 64 B, half a cache line per row: measured (PMC WRITE_SIZE) 110 MB leave the chip for
  // 56.6 MB of C.  So the panel goes out one m-block (16 rows x N) at a time through LDS -- the stages are free now --:
  // every wave drops its columns (+ bias) into a [16][N + 4] buffer, one barrier, then the 512 threads store whole rows,
  // 512 contiguous bytes per half-wave.  Two buffers, so one barrier per m-block.
  constexpr int NCOL = 8 * NBW * 16, CS = NCOL + 4;          // row stride in floats: rows 4 banks apart (conflict-free b128)
  float* cbuf = reinterpret_cast<float*>(lds);
  f32x4 bias[NBW];
#pragma unroll
  for (int n = 0; n < NBW; ++n) {
    bias[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (BIAS) bias[n] = *(const gfloat4*)(uni64(G.bias) + (uint64_t)(min((jb0 + n) * 16, a.N - 16) + 4 * lq) * 4);
  }
  constexpr int RND = NCOL / 128;                             // float4 per row / 32 lanes
  const int srow = tid >> 5, sc4 = tid & 31;                  // this thread's row of the m-block and first float4 of it
  const int nq = a.N >> 2;                                    // float4 per row that exist (the rest is padding: the
  uint32_t scol[RND];                                         //  lanes there store the last real one again)
#pragma unroll
  for (int i = 0; i < RND; ++i) scol[i] = (uint32_t)min(sc4 + 32 * i, nq - 1) * 4;
  const uint64_t cbase = uni64(G.C + (int64_t)bm * a.ldc);
  const uint32_t wofs = (uint32_t)(lc * CS + jb0 * 16 + 4 * lq);
  __syncthreads();                                            // every wave is done with the stages
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    float* cb = cbuf + (m & 1) * 16 * CS;
#pragma unroll
    for (int n = 0; n < NBW; ++n) *reinterpret_cast<f32x4*>(cb + wofs + n * 16) = acc[n][m] + bias[n];
    __syncthreads();
    const uint32_t roff = (uint32_t)(min(m * 16 + srow, rows_here - 1) * a.ldc) * 4;
#pragma unroll
    for (int i = 0; i < RND; ++i) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(cb + srow * CS + scol[i]);
#ifdef PANEL_PROBE_NOCSTORE
      if (v[0] == 123.456f)
#endif
      *(gfloat4*)(cbase + roff + scol[i] * 4) = v;
    }
  }
}

template <int MT, int NBW>
int launch_panel(const PanelArgs& a, int ngroups, bool bias, hipStream_t s) {
  constexpr int stage_bytes = 2 * 3 * MT * 16 * 64 + 512 * 8 + 2 * MT * 16 * 64;     // two stages + the dummy slots (three planes apart)
  constexpr int cbuf_bytes = 2 * 16 * (8 * NBW * 16 + 4) * 4;                          // the epilogue's two row buffers
  constexpr int bytes = stage_bytes > cbuf_bytes ? stage_bytes : cbuf_bytes;
  hipError_t e;
  // more than 64 KB of dynamic LDS needs the attribute, and the attribute is per DEVICE: set once per device and kernel
  // (a process may drive several GPUs: SideStream keeps per-device contexts too)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SCORE_E_BADARG;
  static std::atomic<bool> set_[2][64];
  const int bi = bias ? 1 : 0;
  if (!set_[bi][dev].load(std::memory_order_acquire)) {
    e = bias ? hipFuncSetAttribute((const void*)gemm_panel_kernel<MT, NBW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
             : hipFuncSetAttribute((const void*)gemm_panel_kernel<MT, NBW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    set_[bi][dev].store(true, std::memory_order_release);
  }
  if (bias)
    hipLaunchKernelGGL((gemm_panel_kernel<MT, NBW, true>), dim3(a.tiles * ngroups), dim3(512), bytes, s, a);
  else
    hipLaunchKernelGGL((gemm_panel_kernel<MT, NBW, false>), dim3(a.tiles * ngroups), dim3(512), bytes, s, a);
  e = hipGetLastError();
  return (int)e;
}

int panel_nbw(int N) { return N <= 384 ? 3 : 4; }

}  // namespace

// the panel form pays when its workgroups fill the chip: rows per workgroup (16 * mt) chosen so that the last round of
// 256 is nearly full.  mt = 0: take the tiled kernel.
bool score_gemm_panel_ok(int ngroups, int M, int N, int K, int lda, int ldc, int* mt_out) {
  if (mt_out) *mt_out = 0;
  if (ngroups < 1 || ngroups > 4 || N % 16 || N <= 256 || N > 512 || K % 32 || K < 64 || lda % 4 || ldc % 4 || M < 1) return false;
  int best = 0;
  double best_cost = 1e30;
  for (int mt = 8; mt <= 10; ++mt) {
    const int64_t tiles = (int64_t)ngroups * ((M + 16 * mt - 1) / (16 * mt));
    const double cost = (double)((tiles + 255) / 256) * mt;
    if (cost < best_cost) { best_cost = cost; best = mt; }
  }
  // (rows the chip could have done in the rounds it runs) vs (rows there are): below 0.7 the 128 x 128 tiles fill it better
  if ((double)ngroups * M < 0.7 * best_cost * 16.0 * 256.0) return false;
  if (mt_out) *mt_out = best;
  return true;
}

int64_t score_gemm_panel_image_floats(int N, int K) {
  if (N % 16 || N <= 256 || N > 512 || K % 32) return 0;
  return (int64_t)(K / 32) * (8 * panel_nbw(N)) * 3 * 64 * 4;     // 16 B per lane and fragment
}

int score_gemm_panel_prep(int nimg, const float* const* B, int ldb, int trans, int N, int K, float* const* img, hipStream_t s) {
  if (nimg < 1 || nimg > 4 || score_gemm_panel_image_floats(N, K) == 0) return SCORE_E_SHAPE;
  PrepArgs a;
  for (int i = 0; i < nimg; ++i) { a.B[i] = B[i]; a.img[i] = reinterpret_cast<uint4*>(img[i]); }
  a.ldb = ldb; a.trans = trans; a.N = N; a.K = K; a.NB = 8 * panel_nbw(N);
  const int64_t threads = (int64_t)(K / 32) * a.NB * 64;
  hipLaunchKernelGGL(panel_prep_kernel, dim3((unsigned)((threads + 255) / 256), nimg), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

int score_gemm_panel(int ngroups, const PanelGroup* g, int M, int N, int K, int lda, int ldc, hipStream_t s) {
  int mt = 0;
  if (!score_gemm_panel_ok(ngroups, M, N, K, lda, ldc, &mt)) return SCORE_E_SHAPE;
  PanelArgs a;
  bool bias = g[0].bias != nullptr;
  for (int i = 0; i < ngroups; ++i) {
    a.g[i] = g[i];
    if ((g[i].bias != nullptr) != bias) return SCORE_E_BADARG;
    if (((uintptr_t)g[i].A | (uintptr_t)g[i].C | (uintptr_t)g[i].img | (uintptr_t)g[i].bias) & 15) return SCORE_E_SHAPE;
  }
  const int nbw = panel_nbw(N);
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.NB = 8 * nbw;
  a.tiles = (M + 16 * mt - 1) / (16 * mt);
  if (nbw == 3) {
    if (mt == 8) return launch_panel<8, 3>(a, ngroups, bias, s);
    if (mt == 9) return launch_panel<9, 3>(a, ngroups, bias, s);
    return launch_panel<10, 3>(a, ngroups, bias, s);
  }
  if (mt == 8) return launch_panel<8, 4>(a, ngroups, bias, s);
  if (mt == 9) return launch_panel<9, 4>(a, ngroups, bias, s);
  return launch_panel<10, 4>(a, ngroups, bias, s);
}

// C-ABI ops (include/score_hip.h): the weights' fragment images, the products from prepared images, and both in one call
extern "C" int score_gemm_panel_images(int32_t trans_b, int32_t ngroups, int32_t N, int32_t K, const float* const* Bm, int32_t ldb,
                                       float* images, int64_t image_floats, void* stream) {
  if (!Bm || !images || ngroups < 1 || ngroups > 4 || N <= 0 || K <= 0 || trans_b < 0 || trans_b > 1) return SCORE_E_BADARG;
  const int64_t per = score_gemm_panel_image_floats(N, K);
  if (per == 0) return SCORE_E_SHAPE;
  if (per * ngroups > image_floats) return SCORE_E_WORKSPACE;
  float* img[4];
  for (int i = 0; i < ngroups; ++i) img[i] = images + (int64_t)i * per;
  // Bt(n, k): trans_b = 0 means Bm is [K, N] (C = A . Bm), 1 means Bm is [N, K] (C = A . Bm^T)
  return score_gemm_panel_prep(ngroups, Bm, ldb, trans_b == 0 ? 1 : 0, N, K, img, (hipStream_t)stream);
}

extern "C" int score_gemm_panel_run(int32_t ngroups, int32_t M, int32_t N, int32_t K, const float* const* A, int32_t lda,
                                    float* const* C, int32_t ldc, const float* const* bias, const float* images,
                                    int64_t image_floats, void* stream) {
  if (!A || !C || !images || ngroups < 1 || ngroups > 4 || M <= 0 || N <= 0 || K <= 0) return SCORE_E_BADARG;
  if (!score_gemm_panel_ok(ngroups, M, N, K, lda, ldc, nullptr)) return SCORE_E_SHAPE;
  const int64_t per = score_gemm_panel_image_floats(N, K);
  if (per * ngroups > image_floats) return SCORE_E_WORKSPACE;
  PanelGroup g[4];
  for (int i = 0; i < ngroups; ++i) {
    g[i].A = A[i]; g[i].img = images + (int64_t)i * per; g[i].C = C[i]; g[i].bias = bias ? bias[i] : nullptr;
  }
  return score_gemm_panel(ngroups, g, M, N, K, lda, ldc, (hipStream_t)stream);
}

extern "C" int score_gemm_panel_products(int32_t trans_b, int32_t ngroups, int32_t M, int32_t N, int32_t K, const float* const* A,
                                         int32_t lda, const float* const* Bm, int32_t ldb, float* const* C, int32_t ldc,
                                         const float* const* bias, float* images, int64_t image_floats, void* stream) {
  if (!A || !Bm || !C || !images || ngroups < 1 || ngroups > 4 || M <= 0 || N <= 0 || K <= 0 || trans_b < 0 || trans_b > 1)
    return SCORE_E_BADARG;
  if (!score_gemm_panel_ok(ngroups, M, N, K, lda, ldc, nullptr)) return SCORE_E_SHAPE;
  SCORE_TRY(score_gemm_panel_images(trans_b, ngroups, N, K, Bm, ldb, images, image_floats, stream));
  return score_gemm_panel_run(ngroups, M, N, K, A, lda, C, ldc, bias, images, image_floats, stream);
}
