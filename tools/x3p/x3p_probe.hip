// Round-3 probe (not part of the product library): the fp32-accurate bf16x3 product C[M,N] = A[M,K] . B[N,K]^T with BOTH
// operands arriving as three bf16 planes (x = x0 + x1 + x2 exactly, as gemm_bf16x3.hip splits them) and staged global -> LDS
// by global_load_lds_dwordx4: no VGPR staging, no VALU split, no ds_write.  256 x 128 x 32 tiles, 8 waves (4 x 2, 64 x 64 each),
// one block per CU, two LDS stages (2 x 72 KB), one barrier per k-tile; LDS rows are 64 B (32 k of one plane) with the 16-B
// chunk index XOR-ed with (row >> 2) & 3 on both the source address and the fragment read (conflict-free ds_read_b128).
// tools/x3p_probe.py times it against the shipped kernel on the path's projection shape.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PBM 256
#define PBN 128
#define PBK 32
#define STAGE_BYTES (3 * PBM * 64 + 3 * PBN * 64)      // 73,728

__global__ void x3p_split_kernel(const float* __restrict__ X, int64_t n, unsigned short* __restrict__ p0,
                                 unsigned short* __restrict__ p1, unsigned short* __restrict__ p2) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = X[i];
  const uint32_t xb = __float_as_uint(x);
  const uint32_t h = xb & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(h);
  const uint32_t m = __float_as_uint(r1) & 0xFFFF0000u;
  const float r2 = r1 - __uint_as_float(m);
  p0[i] = (unsigned short)(h >> 16); p1[i] = (unsigned short)(m >> 16); p2[i] = (unsigned short)(__float_as_uint(r2) >> 16);
}

__global__ __launch_bounds__(512, 1) void x3p_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B,
                                                     float* __restrict__ C, int M, int N, int K, int gx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // two stages
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
#ifdef X3P_XCD      // workgroups are dealt round-robin over the 8 XCDs: give every XCD a contiguous run of tiles (x fastest), so
  int wg;          // the column tiles that stream the same A panel share one L2 (common.h: xcd_tile_coords_n)
  {
    const int nwg = (int)gridDim.x, orig = (int)blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
#else
  const int wg = (int)blockIdx.x;
#endif
  const int bx = wg % gx, by = wg / gx;
  const int bm = by * PBM, bn = bx * PBN;
  const int r31 = lane & 31, kh = lane >> 5;
  const int64_t planeA = (int64_t)M * K, planeB = (int64_t)N * K;

  // this wave's nine LDS-DMA pieces of a k-tile (16 rows x 64 B each): piece q = wave * 9 + i
  const unsigned short* src[9];
  uint32_t dst[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int q = wave * 9 + i;
    const bool isA = q < 48;
    const int qq = isA ? q : q - 48;
    const int plane = isA ? qq / 16 : qq / 8, rb = isA ? qq % 16 : qq % 8;
    const int row = rb * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);                        // logical 16-B chunk this lane's LDS slot holds
    const int grow = isA ? min(bm + row, M - 1) : min(bn + row, N - 1);
    src[i] = (isA ? A + plane * planeA : B + plane * planeB) + (int64_t)grow * K + chunk * 8;
    dst[i] = (uint32_t)(isA ? plane * (PBM * 64) + rb * 1024 : 3 * PBM * 64 + plane * (PBN * 64) + rb * 1024);
  }
  auto issue = [&](int k0, int stage) {
#pragma unroll
    for (int i = 0; i < 9; ++i)
#ifndef X3P_NOLOAD
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + k0),
                                       (__attribute__((address_space(3))) void*)(lds + stage * STAGE_BYTES + dst[i]), 16, 0, 0);
#else
      ;
#endif
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment byte offsets inside a stage (per k-step: chunk = ks * 2 + kh, XOR-swizzled by the row)
  uint32_t offA[2][2], offB[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ra = wm * 64 + i * 32 + r31, rbb = wn * 64 + i * 32 + r31;
      offA[i][ks] = (uint32_t)(ra * 64 + (((ks * 2 + kh) ^ ((ra >> 2) & 3)) * 16));
      offB[i][ks] = (uint32_t)(3 * PBM * 64 + rbb * 64 + (((ks * 2 + kh) ^ ((rbb >> 2) & 3)) * 16));
    }

  const int nt = K / PBK;
  auto issue_one = [&](int i, int k0, int stage) {
#ifndef X3P_NOLOAD
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + k0),
                                     (__attribute__((address_space(3))) void*)(lds + stage * STAGE_BYTES + dst[i]), 16, 0, 0);
#endif
  };
  issue(0, 0);
  for (int t = 0; t < nt; ++t) {
    __syncthreads();                       // tile t has landed (the compiler drains the LDS-DMA before the barrier); tile t - 1 is read
#ifndef X3P_SPREAD
    if (t + 1 < nt) issue((t + 1) * PBK, (t + 1) & 1);
#endif
    const unsigned char* st = lds + (t & 1) * STAGE_BYTES;
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
    const bool more = t + 1 < nt;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[2][3], bf[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#ifndef X3P_NOREAD
          af[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + p * (PBM * 64) + offA[i][ks]));
          bf[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + p * (PBN * 64) + offB[i][ks]));
#else
          { uint4 z = make_uint4(offA[i][ks] + t, p, i, ks); af[i][p] = __builtin_bit_cast(bf16x8, z); bf[i][p] = __builtin_bit_cast(bf16x8, z); }
#endif
        }
#pragma unroll
      for (int tt = 0; tt < 6; ++tt)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
#ifndef X3P_NOMFMA
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][TA[tt]], bf[j][TB[tt]], acc[i][j], 0, 0, 0);
#else
            acc[i][j][tt] += (float)af[i][TA[tt]][0] * (float)bf[j][TB[tt]][1];
#endif
#ifdef X3P_SPREAD      // the next tile's nine LDS-DMA pieces dealt over this tile's 48 MFMA slots (one every five) instead of
            {          // back to back in front of them: a piece costs its wave 100 - 185 issue cycles inside a busy phase
              const int g = ks * 24 + tt * 4 + i * 2 + j;
#ifndef X3P_GAP
#define X3P_GAP 5
#endif
              if (more && g % X3P_GAP == X3P_GAP / 2 && g / X3P_GAP < 9) issue_one(g / X3P_GAP, (t + 1) * PBK, (t + 1) & 1);
              __builtin_amdgcn_sched_barrier(0);
            }
#endif
          }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = bn + wn * 64 + j * 32 + r31;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = bm + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < M) C[(int64_t)row * N + col] = acc[i][j][r];
      }
    }
}

extern "C" int x3p_split(const float* X, int64_t n, unsigned short* planes, void* stream) {
  hipLaunchKernelGGL(x3p_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, n, planes,
                     planes + n, planes + 2 * n);
  return (int)hipGetLastError();
}

extern "C" int x3p_gemm(const unsigned short* A, const unsigned short* B, float* C, int M, int N, int K, void* stream) {
  if (K % PBK) return -2;
  const int gx = (N + PBN - 1) / PBN, gy = (M + PBM - 1) / PBM;
  hipFuncSetAttribute((const void*)x3p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
  hipLaunchKernelGGL(x3p_kernel, dim3(gx * gy), dim3(512), 2 * STAGE_BYTES, (hipStream_t)stream, A, B, C, M, N, K, gx);
  return (int)hipGetLastError();
}
