// Profiling aid for head_fused.hip's attention forward kernel (tools/attn_fused_probe.py).
#include "../score_amd/csrc/head_fused.hip"
extern "C" int probe_attn(int B, int T, int H, int NI, const float* q, const float* ur, const float* ir, const float* info,
                          const float* Weff, const float* qz, const float* W4, const float* b4, const float* w5,
                          const float* b5, const int32_t* length, float* inp, float* a1, float* a2, float* score,
                          float* head, int ldh, void* s, int copies, int64_t stride) {
  return score_launch_attn_fwd_fused(B, T, H, NI, 80, 40, q, ur, ir, info, Weff, qz, W4, b4, w5, b5, length, inp, a1, a2,
                                     score, head, ldh, 0, H, (hipStream_t)s, copies, stride);
}
