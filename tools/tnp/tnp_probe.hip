// Round-3 probe (not part of the product library): the recurrences' weight gradients C = [X ; Hm]^T . dY as fp32-accurate
// bf16x3 products on LARGE output tiles, K split into chunks with fixed-order slabs -- the `X^T dY` products that
// gemm_bf16x3.hip runs on 128 x 128 tiles (16 MAC per delivered operand byte: delivery-bound, profiles/r03_probes.md).
//   * a workgroup (8 waves as 2 x 4, one per CU) owns a 192 x (64 * NW) output tile (NW = n-blocks per wave: 4 -> 256
//     columns, 27 MAC / byte; 2 -> 128 columns) and a chunk of K; its partial goes to slab[chunk][M][N].
//   * both operands are k-major in memory ([k][i] rows): a thread loads a 4(k) x 4(i) block (four float4), transposes it in
//     registers, splits each value into three bf16 and writes four k-consecutive values (8 B) per plane and i-row into the LDS
//     image [row][32 k] (64-B rows, 16-B chunks XOR-swizzled) -- the image gemm_panel.hip reads its A fragments from.
//   * one LDS stage (86 KB): barrier, matrix phase, barrier, split + write the next tile (its global loads were issued before
//     the matrix phase).
//   * rows i of the output: [0, Ix) come from X (ld ldx), [Ix, Ix + Ih) from Hm (ld ldh): the [x ; h] row blocks of a TF
//     GRUCell kernel.
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
__device__ __forceinline__ int swz(int row) { return (0x6C >> (((row >> 2) & 3) * 2)) & 3; }

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

constexpr int TM = 192;

template <int NW>
__device__ __forceinline__ void tnp_body(const TnArgs& a, const TnJob& J, unsigned char* lds, int wgid) {
  constexpr int TN = 64 * NW, ROWS = TM + TN, PLANE = ROWS * 64;
  constexpr int NBLK = ROWS / 4 * 8;                 // 4 x 4 blocks of a k-tile: (rows / 4) x (32 / 4)
  constexpr int NRD = (NBLK + 511) / 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = J.Ix + J.Ih;
  const int mt = (M + TM - 1) / TM;
  const int local = wgid - J.first_wg;
  const int tile_m = local % mt, chunk_i = local / mt;
  const int i0 = tile_m * TM;
  const int k0 = chunk_i * J.chunk, k1 = min(a.K, k0 + J.chunk);
  const int nt = (k1 - k0) >> 5;                      // (K and the chunks are multiples of 32)
  const int lc = lane & 15, lq = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;            // wave tile: rows [wm * 96, +96), columns [wn * 16 * NW, ...)

  // staging: block idx = tid + r * 512 -> kb = idx & 7 (k-block of 4), image rows (idx >> 3) * 4 .. + 3
  const float* sp[NRD];
  int sld[NRD];
  uint32_t sdst[NRD];
#pragma unroll
  for (int r = 0; r < NRD; ++r) {
    const int idx = min(tid + r * 512, NBLK - 1);
#ifdef TNP_COALESCED      // lanes run along i (up to 1 KB contiguous per k row and wave instruction) instead of along k
    const int kb = idx / (ROWS / 4), row = (idx % (ROWS / 4)) * 4;
#else
    const int kb = idx & 7, row = (idx >> 3) * 4;
#endif
    if (row < TM) {                                  // A^T image: output rows i0 + row .. + 3
      const int i = min(i0 + row, M - 4);
      if (i < J.Ix) { sp[r] = J.X + i; sld[r] = J.ldx; } else { sp[r] = J.Hm + (i - J.Ix); sld[r] = J.ldh; }
    } else {                                         // B^T image: output columns j0 + (row - TM) .. + 3
      sp[r] = J.Y + J.j0 + min(row - TM, J.N - 4); sld[r] = J.ldy;
    }
    sp[r] += (int64_t)(k0 + kb * 4) * sld[r];
    sdst[r] = (uint32_t)(row * 64 + (((kb >> 1) ^ swz(row)) * 16) + (kb & 1) * 8);    // (rows row .. row + 3 share the swizzle)
  }
  f32x4 sreg[NRD][4];
  auto g_load = [&](int t) {
#pragma unroll
    for (int r = 0; r < NRD; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#ifndef TNP_NOLOAD
        sreg[r][q] = *(const gfloat4*)(sp[r] + (int64_t)(t * 32 + q) * sld[r]);
#else
        sreg[r][q] = f32x4{(float)t, 1.f, (float)q, 2.f};
#endif
  };
  auto s_store = [&]() {
#ifdef TNP_NOSTORE
    if (sreg[0][0][0] != 123.456f) return;
#endif
#pragma unroll
    for (int r = 0; r < NRD; ++r) {
      if (r == NRD - 1 && tid + r * 512 >= NBLK) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c) {                           // image row (row + c): k = kb*4 .. +3 of element c
        uint32_t h[4], m[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) split3(sreg[r][q][c], h[q], m[q], l[q]);
        unsigned char* d = lds + sdst[r] + c * 64;
        *reinterpret_cast<uint2*>(d) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
        *reinterpret_cast<uint2*>(d + PLANE) = make_uint2(pack2(m[0], m[1]), pack2(m[2], m[3]));
        *reinterpret_cast<uint2*>(d + 2 * PLANE) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
      }
    }
  };

  f32x4 acc[NW][6];
#pragma unroll
  for (int n = 0; n < NW; ++n)
#pragma unroll
    for (int m = 0; m < 6; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t aoff = (uint32_t)((wm * 96 + lc) * 64 + ((lq ^ swz(lc)) * 16));
  const uint32_t boff = (uint32_t)((TM + wn * 16 * NW + lc) * 64 + ((lq ^ swz(lc)) * 16));

  g_load(0);
  for (int t = 0; t < nt; ++t) {
    __syncthreads();                    // everybody is done reading the previous tile
    s_store();
    g_load(min(t + 1, nt - 1));
    __syncthreads();
    bf16x8 bf[NW][3];
#pragma unroll
    for (int n = 0; n < NW; ++n)
#pragma unroll
      for (int p = 0; p < 3; ++p) bf[n][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + p * PLANE + boff + n * 1024));
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      bf16x8 af[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) af[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + p * PLANE + aoff + m * 1024));
#pragma unroll
      for (int n = 0; n < NW; ++n) {
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
  for (int m = 0; m < 6; ++m) {
    const int i = i0 + wm * 96 + m * 16 + lc;
#pragma unroll
    for (int n = 0; n < NW; ++n) {
      const int j = wn * 16 * NW + n * 16 + 4 * lq;
#ifdef TNP_NOCSTORE
      if (acc[n][m][0] == 123.456f)
#endif
      if (i < M && j < J.N) *reinterpret_cast<f32x4*>(out + (int64_t)i * J.N + j) = acc[n][m];
    }
  }
}

// one launch for jobs of both widths: N > 128 -> 256-column tiles, else 128-column tiles (wave-uniform per workgroup)
__global__ __launch_bounds__(512, 1) void tnp_kernel(const TnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#ifdef TNP_XCD      // workgroups are dealt round-robin over the 8 XCDs: give every XCD a contiguous run, so that the row tiles of
  int wgid;        // one K-chunk (consecutive ids) share an L2 for the dY rows they all read
  {
    const int nwg = (int)gridDim.x, orig = (int)blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
#else
  const int wgid = (int)blockIdx.x;
#endif
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

// jobs: (X, Hm, Y, slab) per job; all jobs share K; ns / chunk per job from the caller
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
  constexpr int bytes = 3 * (TM + 256) * 64;
  if (hipFuncSetAttribute((const void*)tnp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -3;
  hipLaunchKernelGGL(tnp_kernel, dim3(wg), dim3(512), bytes, s, a);
  return (int)hipGetLastError();
}

extern "C" int tnp_reduce(const float* slab, int ns, int64_t n, float* C, void* stream) {
  hipLaunchKernelGGL(tnp_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, slab, ns, n, C);
  return (int)hipGetLastError();
}
