// Profiling aid for gemm_bf16x3.hip: the kernel compiled with a probe macro (see tools/x3_probe.py).
#include "../score_amd/csrc/gemm_bf16x3.hip"
extern "C" int probe_launch(int trans, int wm, int gx, int gy, int gz, int M, int N, int K, const float* A, int lda,
                            const float* B, int ldb, float* C, int ldc, int kc, float* slab, void* s) {
  GemmGroup g;
  g.n = 1;
  GemmProb& p = g.p[0];
  p.A = A; p.B = B; p.C = C; p.slab = slab; p.bias = nullptr; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.k_chunk = kc; p.gx = gx; p.gy = gy; p.nblocks = gx * gy * gz;
  g.total_blocks = (p.nblocks + 7) & ~7;
  return score_launch_gemm_bf16x3(trans, wm, g, nullptr, 0, 1.f, nullptr, 0, (hipStream_t)s);
}
