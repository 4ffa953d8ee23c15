// Profiling aid for gemm_bf16x3.hip: the kernel compiled with a probe macro (see tools/x3_probe.py).
#include "../score_amd/csrc/gemm_bf16x3.hip"
extern "C" int probe_launch(int trans, int wm, int gx, int gy, int gz, int M, int N, int K, const float* A, int lda,
                            const float* B, int ldb, float* C, int ldc, int kc, float* slab, void* s) {
  return score_launch_gemm_bf16x3(trans, wm, dim3(gx, gy, gz), M, N, K, A, lda, B, ldb, C, ldc, nullptr, 0, 1.f,
                                  nullptr, 0, kc, slab, (hipStream_t)s);
}
