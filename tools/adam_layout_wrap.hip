// Layout probe for the table optimizer's row kernels (tools/adam_layout_probe.py): the rows with a gradient of a step
// (every ~9th row of 1.53 M at cfg-3) read p, m, v, g and write p, m, v -- as four separate [N][D] arrays (shipped) or
// as one interleaved [N][4][D] block.  Same arithmetic, timing only.
#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4g(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void upd(float4& p, float4& m, float4& v, float4 g) {
  m.x = fmaf(g.x - m.x, 0.1f, m.x); m.y = fmaf(g.y - m.y, 0.1f, m.y); m.z = fmaf(g.z - m.z, 0.1f, m.z); m.w = fmaf(g.w - m.w, 0.1f, m.w);
  v.x = fmaf(g.x * g.x - v.x, 0.001f, v.x); v.y = fmaf(g.y * g.y - v.y, 0.001f, v.y); v.z = fmaf(g.z * g.z - v.z, 0.001f, v.z); v.w = fmaf(g.w * g.w - v.w, 0.001f, v.w);
  p.x -= m.x * 0.01f * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v.x) + 1e-8f); p.y -= m.y * 0.01f * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v.y) + 1e-8f);
  p.z -= m.z * 0.01f * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v.z) + 1e-8f); p.w -= m.w * 0.01f * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v.w) + 1e-8f);
}
// one 16-lane group per row of `rows` (D = 64)
__global__ __launch_bounds__(256) void rows_split(float* p, float* m, float* v, const float* g, const int* rows, int n) {
  const int64_t gi = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  if (gi >= n) return;
  const int64_t e = (int64_t)rows[gi] * 64 + (threadIdx.x & 15) * 4;
  float4 P = ld4g(p + e), M = ld4g(m + e), V = ld4g(v + e), G = ld4g(g + e);
  upd(P, M, V, G);
  st4g(p + e, P); st4g(m + e, M); st4g(v + e, V);
}
// ... plus the per-row state the shipped kernel also writes: a state byte and a step counter
__global__ __launch_bounds__(256) void rows_split_state(float* p, float* m, float* v, const float* g, const int* rows, int n,
                                                        unsigned char* flags, unsigned int* step) {
  const int64_t gi = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  if (gi >= n) return;
  const int r = rows[gi];
  const int64_t e = (int64_t)r * 64 + (threadIdx.x & 15) * 4;
  float4 P = ld4g(p + e), M = ld4g(m + e), V = ld4g(v + e), G = ld4g(g + e);
  upd(P, M, V, G);
  st4g(p + e, P); st4g(m + e, M); st4g(v + e, V);
  if ((threadIdx.x & 15) == 0) { flags[r] = 1; step[r] = 7u; }
}
__global__ __launch_bounds__(256) void rows_inter(float* t, const int* rows, int n) {
  const int64_t gi = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  if (gi >= n) return;
  float* b = t + (int64_t)rows[gi] * 256 + (threadIdx.x & 15) * 4;
  float4 P = ld4g(b), M = ld4g(b + 64), V = ld4g(b + 128), G = ld4g(b + 192);
  upd(P, M, V, G);
  st4g(b, P); st4g(b + 64, M); st4g(b + 128, V);
}
// the forward's side of the trade: a gather of p rows only (256 B out of every 1 KB when interleaved)
__global__ __launch_bounds__(256) void gather_p(const float* t, int64_t ld, const int* rows, int n, float* out) {
  const int64_t gi = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  if (gi >= n) return;
  st4g(out + gi * 64 + (threadIdx.x & 15) * 4, ld4g(t + (int64_t)rows[gi] * ld + (threadIdx.x & 15) * 4));
}
extern "C" int probe(int which, float* a, float* b, float* c, float* d, const int* rows, int n, float* out, void* s) {
  const unsigned blocks = (unsigned)(((int64_t)n * 16 + 255) / 256);
  if (which == 0) hipLaunchKernelGGL(rows_split, dim3(blocks), dim3(256), 0, (hipStream_t)s, a, b, c, d, rows, n);
  else if (which == 1) hipLaunchKernelGGL(rows_inter, dim3(blocks), dim3(256), 0, (hipStream_t)s, a, rows, n);
  else if (which == 4) hipLaunchKernelGGL(rows_split_state, dim3(blocks), dim3(256), 0, (hipStream_t)s, a, b, c, d, rows, n,
                                          reinterpret_cast<unsigned char*>(out), reinterpret_cast<unsigned int*>(out) + (1 << 20));
  else if (which == 2) hipLaunchKernelGGL(gather_p, dim3(blocks), dim3(256), 0, (hipStream_t)s, a, (int64_t)64, rows, n, out);
  else hipLaunchKernelGGL(gather_p, dim3(blocks), dim3(256), 0, (hipStream_t)s, a, (int64_t)256, rows, n, out);
  return (int)hipGetLastError();
}
