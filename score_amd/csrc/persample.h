// Whole-model kernels for the reference's own shapes (train_score.py:15-16, 285-372: D = 16, H = 32, B = 100 / 200):
// ONE workgroup per sample runs the whole forward pass of score.py:188-224 (gather -> co-attention -> GRU input
// projections -> both recurrences -> temporal attention -> build_fc_net -> loss term) in one launch, and its mirror runs
// the backward pass down to everything the weight-gradient products and the row scatter consume.  Samples are independent
// through the whole graph (bn1 is an inference-mode affine, score.py:69), so no workgroup ever waits for another.
// At these shapes the layer-by-layer sequence is ~25 dependent kernels of 5 - 15 us and ~55 launches per step: the step
// is bound by launch latency on both sides of the queue, not by bytes or flops.
//
// Weights reach the matrix cores as "images": W[K][N] cut into 16-column tiles and 16-deep k chunks, each (tile, chunk) a
// 1-KB block of 64 float4 -- lane l = (lq = l / 16, lc = l % 16) holds W[16c + 4 lq + s][16 ct + lc], s = 0..3 -- i.e. the
// B operands of four consecutive v_mfma_f32_16x16x4_f32 steps of a lane are ONE coalesced 16-byte load.  The matching A
// operand is one ds_read_b128 of the activation row (k runs dealt to the lane quarters: the sum over k does not care).
// ps_prep_kernel writes the images once per step from the live variables (beside nothing: it is the step's first launch).
#pragma once
#include "common.h"

#define PS_NT 512          // threads per workgroup (8 waves)
#define PS_NW 8
#define PS_MAX_B 512       // batch sizes above this take the layer-by-layer path (weights would be streamed B times)
#define SCORE_PS_MAX_DEVICES 16
#define PS_MAX_PADS 40

// floats of the image of a [K][N] matrix
static inline int64_t ps_image_floats(int K, int N) { return (int64_t)((K + 15) / 16) * ((N + 15) / 16) * 256; }

enum { PS_SRC_PLAIN = 0, PS_SRC_WXCAT = 1, PS_SRC_WEFF = 2, PS_SRC_WQ = 3 };
// one matrix to image: logical element (k, n) = trans ? src(n, k) : src(k, n), src by `kind`:
//   PLAIN  W[r * ld + c]
//   WXCAT  r-th x row of [gates/kernel | candidate/kernel] of a GRU: c < 2H ? W[r * 2H + c] : W2[r * H + c - 2H]
//   WEFF   folded dense_3 (head.hip): r < Dk ? W[(Dk + r) * ld + c] - W[(2Dk + r) * ld + c] : W[(3Dk + r - Dk) * ld + c]
//   WQ     W[r * ld + c] + W[(2Dk + r) * ld + c]
struct PsImgJob { const float* W; const float* W2; float* img; int K, N, ld, kind, trans, aux /*H or Dk*/, first_block, pad; };
#define PS_MAX_IMG 16
struct PsPrepArgs {
  PsImgJob job[PS_MAX_IMG]; int njobs, img_blocks;
  const float* wreg; int64_t n_reg; float* part;        // L2 partial sums (256 blocks); part[256] (a word) <- 0: PsFwdArgs.done
  float* zero; int64_t zero_floats; int zero_blocks;    // dense gradient buffer cleared for the backward pass
};

struct PsShape {
  int B, A, Tidx, K, D4, Fu, Fi, H, Du, Di, I, NI, Dk, Dhead, off_u, off_i, off_ti, off_tu, MP, Bglobal;
  int GS[2], nslots[2], V0, Vtot;        // gather geometry (lanes of a (slice, call) group; call 0 padded to whole waves)
};

// float offsets of the images inside the image region
struct PsImages {
  int64_t wx[2], q2, wq, weff, w4, fc1, fc2;              // forward:  [K][N] as the layers use them
  int64_t fc2t, fc1t, w4t, wefft, wqt, q2t, wxt[2];       // backward: the transposes
  int64_t total;
};

struct PsFwdArgs {
  PsShape s;
  const int32_t* idx1[2]; const int32_t* idx2[2]; const int32_t* tu; const int32_t* ti; const int32_t* label; const int32_t* length;
  const float* table; uint32_t n_rows; int32_t* id_status;
  const float* W;                        // flat dense variables
  int64_t ca_w[2], ca_b[2], gk[2], gb[2], ck[2], cb[2], at_b[4], at_w5, bn_g, bn_b, fc_b[3], fc_w3;
  const float* img; PsImages im;
  float* query; float* head_inp; float* xside[2]; float* info; float* rsave[2]; float* gates[2]; float* gru_out[2];
  float* gru_final[2]; float* q; float* ainp; float* a1; float* a2; float* att_score; float* bn; float* f1; float* f2;
  float* logit; float* y; float* lossb; float* dlogit; float* dz2;
  float keep, rs; int drop; const uint8_t* mask0; const uint8_t* mask1; uint64_t seed0, seed1; const uint64_t* seed_dev;
  // the loss of the step (score.py:74-81, 91-94), reduced by the LAST workgroup to finish (round 5: it was a one-workgroup launch
  // of its own on the side stream -- a launch, a fork and two event calls per step): `done` counts finished workgroups (zeroed by
  // ps_prep_kernel), part = the 256 partial sums of squares of the regularised range, loss_host = optional pinned host copy
  float* loss; float* loss_host; const float* part; unsigned int* done; float lambda, inv_bglobal;
};

struct PsBwdArgs {
  PsShape s;
  const int32_t* idx1[2]; const int32_t* idx2[2]; const int32_t* length;
  const float* table; uint32_t n_rows;
  const float* W;
  int64_t ca_w[2], gk[2], ck[2], at_w5, bn_g;
  const float* img; PsImages im;
  // saved by the forward pass
  const float* query; const float* head_inp; const float* info; const float* rsave[2]; const float* gates[2];
  const float* gru_out[2]; const float* q; const float* ainp; const float* a1; const float* a2; const float* att_score;
  const float* f1; const float* dz2;
  // what the weight-gradient products / column sums and the row scatter read
  float* dz1; float* dbn; float* dgstage; float* ds; float* da2; float* da1; float* adzsum; float* dq;
  float* dxproj[2]; float* rh[2]; float* hprev[2]; float* dxside[2]; float* pcoef[2]; float* dzcoef[2]; float* dtgt;
  float* S; float* caslab[2];            // S [2][B]; co-attention dW1 | dW2 partial of this sample: caslab[c][b][2 Dx]
  float keep, rs;
  // the alignment padding between the tensors of the flat dense gradient: nothing of the pass writes it, ApplyAdam and the L2
  // norm run over it -- workgroup 0 clears it (every other float of grad_w is overwritten by the pass: no memset launch)
  float* gw; int npad; int pad_off[PS_MAX_PADS]; int pad_len[PS_MAX_PADS];
};

int ps_plan_shape(int B, int A, int Tidx, int K, int D, int Fu, int Fi, int H, int NI, int Dk, int Dhead, int off_u, int off_i,
                  int off_ti, int off_tu, int Bglobal, PsShape* out);       // SCORE_E_SHAPE: not covered
void ps_plan_images(const PsShape& s, PsImages* im);
int score_launch_ps_prep(const PsPrepArgs& a, hipStream_t s);
int score_launch_ps_fwd(const PsFwdArgs& a, hipStream_t s);
int score_launch_ps_bwd(const PsBwdArgs& a, hipStream_t s);
