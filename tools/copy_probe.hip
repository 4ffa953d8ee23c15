// Variants of a float4 stream copy: which launch shape reaches the guide's ~6.3 TB/s on this pool's boxes?
// (bench.py's roofline.peak_measured uses the winner: score_stream_copy in csrc/head.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_k(v4f* __restrict__ d, const v4f* __restrict__ s, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    v4f r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(r[u], d + i + u * stride); else d[i + u * stride] = r[u]; }
  }
  for (; i < n4; i += stride) d[i] = s[i];
}
// contiguous chunk per block: block b owns [b*chunk, (b+1)*chunk)
template <int U>
__global__ __launch_bounds__(256) void copy_chunk(v4f* __restrict__ d, const v4f* __restrict__ s, int64_t n4, int64_t chunk) {
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n4 ? lo + chunk : n4;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256 * U) {
    v4f r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = (i + u * 256 < hi) ? s[i + u * 256] : v4f{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < hi) d[i + u * 256] = r[u];
  }
}
extern "C" int probe(int which, int blocks, float* d, const float* s, int64_t n, void* st) {
  const int64_t n4 = n / 4;
  hipStream_t q = (hipStream_t)st;
  v4f* D = (v4f*)d; const v4f* S = (const v4f*)s;
  switch (which) {
    case 0: hipLaunchKernelGGL((copy_k<1, false>), dim3(blocks), dim3(256), 0, q, D, S, n4); break;
    case 1: hipLaunchKernelGGL((copy_k<4, false>), dim3(blocks), dim3(256), 0, q, D, S, n4); break;
    case 2: hipLaunchKernelGGL((copy_k<8, false>), dim3(blocks), dim3(256), 0, q, D, S, n4); break;
    case 3: hipLaunchKernelGGL((copy_k<4, true>), dim3(blocks), dim3(256), 0, q, D, S, n4); break;
    case 4: hipLaunchKernelGGL((copy_k<8, true>), dim3(blocks), dim3(256), 0, q, D, S, n4); break;
    case 5: hipLaunchKernelGGL((copy_chunk<4>), dim3(blocks), dim3(256), 0, q, D, S, n4, (n4 + blocks - 1) / blocks); break;
    case 6: hipLaunchKernelGGL((copy_chunk<8>), dim3(blocks), dim3(256), 0, q, D, S, n4, (n4 + blocks - 1) / blocks); break;
    case 7: return (int)hipMemcpyAsync(d, s, n * 4, hipMemcpyDeviceToDevice, q);
  }
  return (int)hipGetLastError();
}
