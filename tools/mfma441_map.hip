// Lane mapping of v_mfma_f32_4x4x1_16b_f32 (tools/mfma441_map.py prints it).
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 1.0f, acc, 0, 0, 0);          // D_b[i][j] = A_b[i]
  for (int r = 0; r < 4; ++r) out[l * 8 + r] = acc[r];
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
  acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)l, acc2, 0, 0, 0);        // D_b[i][j] = B_b[j]
  for (int r = 0; r < 4; ++r) out[l * 8 + 4 + r] = acc2[r];
}
extern "C" int run(float* out, void* s) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, (hipStream_t)s, out); return (int)hipGetLastError(); }
