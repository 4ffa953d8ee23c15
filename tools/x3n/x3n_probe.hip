// Profiling aid for csrc/gemm_panel.hip: the product kernel compiled with one ingredient stripped at a time
// (-DPANEL_PROBE_NOMFMA / NOALOAD / NOASTORE / NOBLOAD / NOREAD / NOCSTORE: wrong results, timing only) and an explicit
// choice of rows per panel (mt) -- see tools/x3n_probe.py, tools/x3n_strip.sh.
#include "../../score_amd/csrc/gemm_panel.hip"

extern "C" int64_t x3n_image_bytes(int N, int K) { return score_gemm_panel_image_floats(N, K) * 4; }

extern "C" int x3n_prep(int nimg, const float* const* B, int ldb, int trans, int N, int K, float* const* img, void* stream) {
  return score_gemm_panel_prep(nimg, B, ldb, trans, N, K, img, (hipStream_t)stream);
}

// groups share M, N, K, lda, ldc; A / image / C / bias per group
extern "C" int x3n_gemm(int ngroups, const float* const* A, const float* const* img, float* const* C, const float* const* bias, int M,
                        int N, int K, int lda, int ldc, int mt, void* stream) {
  if (ngroups < 1 || ngroups > 4 || score_gemm_panel_image_floats(N, K) == 0) return -2;
  PanelArgs a;
  for (int g = 0; g < ngroups; ++g) { a.g[g].A = A[g]; a.g[g].img = img[g]; a.g[g].C = C[g]; a.g[g].bias = bias ? bias[g] : nullptr; }
  const int nbw = panel_nbw(N);
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.NB = 8 * nbw;
  a.tiles = (M + 16 * mt - 1) / (16 * mt);
  hipStream_t s = (hipStream_t)stream;
  const bool b = bias != nullptr;
  if (nbw == 3 && mt == 8) return launch_panel<8, 3>(a, ngroups, b, s);
  if (nbw == 3 && mt == 9) return launch_panel<9, 3>(a, ngroups, b, s);
  if (nbw == 3 && mt == 10) return launch_panel<10, 3>(a, ngroups, b, s);
  if (nbw == 4 && mt == 8) return launch_panel<8, 4>(a, ngroups, b, s);
  if (nbw == 4 && mt == 9) return launch_panel<9, 4>(a, ngroups, b, s);
  if (nbw == 4 && mt == 10) return launch_panel<10, 4>(a, ngroups, b, s);
  return -2;
}
