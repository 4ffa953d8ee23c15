// Round-3 probe, second form (tools/tnp_probe.py -> TNP2=1): the recurrences' weight gradients C = [X ; Hm]^T . dY on
// 144 x 256 (or 144 x 128) output tiles with TWO LDS stages, coalesced staging and an LDS image both the staging writes and
// the fragment reads reach without bank conflicts.
//   * workgroup = 8 waves, one per CU; a wave owns all 144 rows (9 m-blocks) x 32 (16) columns; K in chunks, partials to
//     slab[chunk][M][N] (fixed order: reproducible).
//   * staging: a thread loads a 4(k) x 4(i) block -- four float4, lanes running along i: up to 1 KB contiguous per k row and wave
//     instruction --, transposes it in registers, splits each value into three bf16 and writes, per plane, four 8-byte cells
//     (4 consecutive k of one row).
//   * LDS image of a stage: cell(plane p, k-block kb = 0..7, row r) at p*PL + kb*SK + (r & 3)*SC + (r >> 2)*8.  Lanes of a
//     staging instruction write consecutive cells (conflict-free ds_write_b64); a 16x16x32 fragment (row r0 + lane % 16,
//     k-blocks 2*(lane / 16) and + 1) is two ds_read_b64, conflict-free with SC = 64 (mod 256) and 2*SK = 32 (mod 256).
//   * every XCD gets a contiguous run of workgroups: the four row tiles of a K-chunk share an L2 for their dY rows.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4 gfloat4;

__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  const uint32_t xb = __float_as_uint(x);
  h = xb & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xFFFF0000u;
  l = __float_as_uint(r1 - __uint_as_float(m));
}
__device__ __forceinline__ uint32_t pack2(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
__device__ __forceinline__ uint64_t uni64(const void* p_) {
  const uint64_t p = reinterpret_cast<uint64_t>(p_);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
  return ((uint64_t)hi << 32) | lo;
}

#if defined(TNP_NOMFMA)
#define T_MFMA(a, b, c) (c)
#else
#define T_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif

struct TnJob {
  const float* X; const float* Hm; const float* Y;   // X [K, Ix] (ldx), Hm [K, Ih] (ldh), Y [K, >= j0 + N] (ldy)
  float* slab;                                       // [ns][M][N], M = Ix + Ih
  int ldx, ldh, ldy, Ix, Ih, j0, N, ns, chunk, first_wg;   // chunk: rows of K per workgroup (multiple of 32)
};
struct TnArgs { TnJob j[8]; int njobs, K; };

#ifndef TNP_TM
#define TNP_TM 128
#endif
constexpr int TM = TNP_TM, MB = TM / 32;      // 8 waves as 2 (rows) x 4 (columns): MB m-blocks per wave

template <int NBW>        // n-blocks per wave: 4 -> 256-column tiles, 2 -> 128
struct Img {
  static constexpr int TN = 64 * NBW, R = TM + TN, NT4 = R / 4;
  static constexpr int SC = NT4 * 16 + ((64 - (NT4 * 16) % 256 + 256) % 256);     // = 64 (mod 256)
  static constexpr int SK = 4 * SC + ((256 - (4 * SC) % 256) % 256);              // = 0 (mod 256)
  static constexpr int PL = 4 * SK, STAGE = 3 * PL;
  static constexpr int NBLK = NT4 * 4;                                            // 8(k) x 4(i) blocks of a k-tile
  static_assert(NBLK <= 512, "one staging round");
};

template <int NBW>
__device__ __forceinline__ void tnp_body(const TnArgs& a, const TnJob& J, unsigned char* lds, int wgid) {
  typedef Img<NBW> G;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = J.Ix + J.Ih;
  const int mt = (M + TM - 1) / TM;
  const int local = wgid - J.first_wg;
  const int tile_m = local % mt, chunk_i = local / mt;
  const int i0 = tile_m * TM;
  const int k0 = chunk_i * J.chunk, k1 = min(a.K, k0 + J.chunk);
  const int nt = (k1 - k0) >> 5;                      // (K and the chunks are multiples of 32)
  const int lc = lane & 15, lq = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;

  // staging: block idx = tid -> t = idx % NT4 (rows 4t .. 4t+3 of the image), kq = idx / NT4 (k = 8 kq .. + 7); threads past
  // the last block repeat it (loads and stores: the same values to the same cells)
  uint64_t sp;
  uint32_t sld, sdst;
  {
    const int idx = min(tid, G::NBLK - 1);
    const int kq = idx / G::NT4, t = idx - kq * G::NT4;
    const int row = t * 4;
    const float* p;
    if (row < TM) {                                  // A^T rows: output rows i0 + row .. + 3
      const int i = min(i0 + row, M - 4);
      if (i < J.Ix) { p = J.X + i; sld = (uint32_t)J.ldx * 4; } else { p = J.Hm + (i - J.Ix); sld = (uint32_t)J.ldh * 4; }
    } else {                                         // B^T rows: output columns j0 + (row - TM) .. + 3
      p = J.Y + J.j0 + min(row - TM, J.N - 4); sld = (uint32_t)J.ldy * 4;
    }
    sp = reinterpret_cast<uint64_t>(p) + (uint64_t)(k0 + kq * 8) * sld;
    sdst = (uint32_t)(kq * G::SK + t * 16);
  }
  f32x4 sreg[8];
  auto g_load = [&](int t) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
#ifndef TNP_NOLOAD
      sreg[q] = *(const gfloat4*)(sp + (uint64_t)(t * 32 + q) * sld);
#else
      sreg[q] = f32x4{(float)t, 1.f, (float)q, 2.f};
#endif
  };
  auto s_store = [&](int stage) {
#ifdef TNP_NOSTORE
    if (sreg[0][0] != 123.456f) return;
#endif
    unsigned char* d0 = lds + stage * G::STAGE + sdst;
#pragma unroll
    for (int c = 0; c < 4; ++c) {                           // image row 4t + c: k = 8 kq .. + 7 of element c, one 16-B cell a plane
      uint32_t h[8], m[8], l[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) split3(sreg[q][c], h[q], m[q], l[q]);
      unsigned char* d = d0 + c * G::SC;
      *reinterpret_cast<u32x4*>(d) = u32x4{pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7])};
      *reinterpret_cast<u32x4*>(d + G::PL) = u32x4{pack2(m[0], m[1]), pack2(m[2], m[3]), pack2(m[4], m[5]), pack2(m[6], m[7])};
      *reinterpret_cast<u32x4*>(d + 2 * G::PL) = u32x4{pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7])};
    }
  };

  f32x4 acc[NBW][MB];
#pragma unroll
  for (int n = 0; n < NBW; ++n)
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment reads: cell(kq = lq, row r0 + lc), one ds_read_b128; r0 = 16 (wm MB + m) (A) or TM + 16 (wn NBW + n) (B)
  const uint32_t fbase = (uint32_t)(lq * G::SK + (lc & 3) * G::SC + (lc >> 2) * 16);
  const uint32_t aoff = fbase + (uint32_t)(wm * MB * 64);
  const uint32_t boff = fbase + (uint32_t)((TM / 4 + 4 * wn * NBW) * 16);
  auto frag = [&](const unsigned char* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p)); };

  g_load(0);
  s_store(0);
  g_load(min(1, nt - 1));
  for (int t = 0; t < nt; ++t) {
    __syncthreads();                    // stage t & 1 is complete; nobody reads stage (t + 1) & 1 any more
    s_store((t + 1) & 1);
    g_load(min(t + 2, nt - 1));
    uint32_t so = (uint32_t)(t & 1) * G::STAGE;
    const unsigned char* st = lds + so;
    bf16x8 bf[NBW][3];
#pragma unroll
    for (int n = 0; n < NBW; ++n)
#pragma unroll
      for (int p = 0; p < 3; ++p) bf[n][p] = frag(st + p * G::PL + boff + n * 64);
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      bf16x8 af[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) af[p] = frag(st + p * G::PL + aoff + m * 64);
#pragma unroll
      for (int n = 0; n < NBW; ++n) {
        f32x4 c = acc[n][m];
        c = T_MFMA(bf[n][2], af[0], c);
        c = T_MFMA(bf[n][0], af[2], c);
        c = T_MFMA(bf[n][1], af[1], c);
        c = T_MFMA(bf[n][1], af[0], c);
        c = T_MFMA(bf[n][0], af[1], c);
        c = T_MFMA(bf[n][0], af[0], c);
        acc[n][m] = c;
      }
    }
  }
  // partial tile -> slab[chunk_i][M][N]: lane = output row (i) lc of the m-block, four consecutive columns
  float* out = J.slab + (int64_t)chunk_i * M * J.N;
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    const int i = i0 + (wm * MB + m) * 16 + lc;
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
      const int j = (wn * NBW + n) * 16 + 4 * lq;
#ifdef TNP_NOCSTORE
      if (acc[n][m][0] == 123.456f)
#endif
      if (i < M && j < J.N) *reinterpret_cast<f32x4*>(out + (int64_t)i * J.N + j) = acc[n][m];
    }
  }
}

__global__ __launch_bounds__(512, 1) void tnp_kernel(const TnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  int wgid;        // every XCD a contiguous run of workgroups (they are dealt round-robin over the 8 XCDs)
  {
    const int nwg = (int)gridDim.x, orig = (int)blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  int ji = 0;
  while (ji + 1 < a.njobs && wgid >= a.j[ji + 1].first_wg) ++ji;
  const TnJob& J = a.j[ji];
  if (J.N > 128) tnp_body<4>(a, J, lds, wgid); else tnp_body<2>(a, J, lds, wgid);
}

__global__ void tnp_reduce_kernel(const float* __restrict__ slab, int ns, int64_t n, float* __restrict__ C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int z = 0; z < ns; ++z) s += slab[(int64_t)z * n + i];
  C[i] = s;
}

extern "C" int tnp_launch(int njobs, const float* const* X, const float* const* Hm, const float* const* Y, float* const* slab,
                          const int* ldx, const int* ldh, const int* ldy, const int* Ix, const int* Ih, const int* j0, const int* N,
                          const int* ns, const int* chunk, int K, int nw, void* stream) {
  if (njobs < 1 || njobs > 8 || K % 32) return -2;
  TnArgs a;
  a.njobs = njobs; a.K = K;
  int wg = 0;
  for (int i = 0; i < njobs; ++i) {
    TnJob& j = a.j[i];
    j.X = X[i]; j.Hm = Hm[i]; j.Y = Y[i]; j.slab = slab[i]; j.ldx = ldx[i]; j.ldh = ldh[i]; j.ldy = ldy[i]; j.Ix = Ix[i]; j.Ih = Ih[i];
    j.j0 = j0[i]; j.N = N[i]; j.ns = ns[i]; j.chunk = chunk[i]; j.first_wg = wg;
    if (chunk[i] % 32 || N[i] > 256 || N[i] % 4 || (Ix[i] % 4) || (Ih[i] % 4)) return -2;
    wg += ((Ix[i] + Ih[i] + TM - 1) / TM) * ns[i];
  }
  hipStream_t s = (hipStream_t)stream;
  constexpr int bytes = 2 * Img<4>::STAGE;
  static_assert(bytes <= 160 * 1024, "LDS");
  if (hipFuncSetAttribute((const void*)tnp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -3;
  hipLaunchKernelGGL(tnp_kernel, dim3(wg), dim3(512), bytes, s, a);
  return (int)hipGetLastError();
}

extern "C" int tnp_reduce(const float* slab, int ns, int64_t n, float* C, void* stream) {
  hipLaunchKernelGGL(tnp_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, slab, ns, n, C);
  return (int)hipGetLastError();
}
