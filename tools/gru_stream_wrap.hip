// Profiling aid for gru_stream.hip: the kernels compiled with a probe macro (see tools/gru_stream_probe.py).
#include "../score_amd/csrc/gru_stream.hip"
#include <string.h>
extern "C" int probe_gru(int dir, int B, int T, int H, const float* xproj, const float* Wg, const float* Wc,
                         const int32_t* length, float* out, float* gates, const float* dout, float* dxproj, float* rh,
                         float* hprev, float* tmp, int64_t tmp_floats, void* s) {
  GruArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.T = T; a.H = H; a.length = length; a.tmp = tmp; a.tmp_floats = tmp_floats;
  for (int i = 0; i < 2; ++i) {
    GruSide& g = a.s[i];
    const int64_t o = (int64_t)i * B * T;
    g.xproj = xproj + o * 3 * H; g.Wg = Wg + (int64_t)i * H * 2 * H; g.ldwg = 2 * H; g.Wc = Wc + (int64_t)i * H * H; g.ldwc = H;
    g.out = out + o * H; g.ldo = H; g.gates = gates + o * 3 * H; g.dout = dout + o * H; g.lddo = H;
    g.dxproj = dxproj + o * 3 * H; g.rh = rh + o * H; g.hprev = hprev + o * H;
  }
  return dir == 0 ? score_gru_fwd_stream(a, 2, (hipStream_t)s) : score_gru_bwd_stream(a, 2, (hipStream_t)s);
}
