// Profiling aid for gru.hip's register-resident kernels (see tools/gru_reg_probe.py).
#include "../score_amd/csrc/gru.hip"
bool score_gru_stream_ok(int) { return false; }
int64_t score_gru_stream_tmp_floats(int, int) { return 0; }
int score_gru_fwd_stream(GruArgs&, int, hipStream_t) { return SCORE_E_SHAPE; }
int score_gru_bwd_stream(GruArgs&, int, hipStream_t) { return SCORE_E_SHAPE; }
bool score_gru_x3_ok(int, int) { return false; }
int score_gru_fwd_x3(GruArgs&, int, hipStream_t) { return SCORE_E_SHAPE; }
int score_gru_bwd_x3(GruArgs&, int, hipStream_t) { return SCORE_E_SHAPE; }
int score_gemm_same_shape(int, int, int, int, int, const float* const*, int, const float* const*, int, float* const*, int, int, int,
                          float*, int64_t, hipStream_t, const float* const*) { return SCORE_E_SHAPE; }
extern "C" int probe_gru(int dir, int B, int T, int H, const float* xproj, const float* Wg, const float* Wc,
                         const int32_t* length, float* out, float* gates, const float* dout, float* dxproj, float* rh,
                         float* hprev, void* s) {
  GruArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.T = T; a.H = H; a.length = length; a.nw8 = 1;
  for (int i = 0; i < 2; ++i) {
    GruSide& g = a.s[i];
    const int64_t o = (int64_t)i * B * T;
    g.xproj = xproj + o * 3 * H; g.Wg = Wg + (int64_t)i * H * 2 * H; g.ldwg = 2 * H; g.Wc = Wc + (int64_t)i * H * H; g.ldwc = H;
    g.out = out + o * H; g.ldo = H; g.gates = gates + o * 3 * H; g.dout = dout + o * H; g.lddo = H;
    g.dxproj = dxproj + o * 3 * H; g.rh = rh + o * H; g.hprev = hprev + o * H;
  }
  return dir == 0 ? score_gru_fwd_multi(a, 2, (hipStream_t)s) : score_gru_bwd_multi(a, 2, (hipStream_t)s);
}
