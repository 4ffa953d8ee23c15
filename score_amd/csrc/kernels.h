// Internal (non-ABI) launchers shared between the translation units of libscore_hip.
#pragma once
#include "common.h"

// gemm.hip / gemm_bf16x3.hip
// One launch can carry several GEMM problems of the same operand layout and tile shape (the independent
// weight-gradient products of a backward pass): a workgroup finds its problem from the running block count.
// Every problem's block range starts at a multiple of 8, so (index inside the problem) % 8 still names the
// XCD group of common.h's tile order; the padding blocks exit at once.
#define GEMM_GROUP_MAX 16
struct GemmProb {
  const float* A; const float* B; float* C; float* slab;   // slab: split-K partials [gz][M][N] (null: write C)
  const float* bias;                                       // this problem's own bias (null: the launch's)
  int M, N, K, lda, ldb, ldc, k_chunk, gx, gy, nblocks;    // nblocks = gx*gy*gz; the launch gives it align8(nblocks)
};
struct GemmGroup { int n; int total_blocks; GemmProb p[GEMM_GROUP_MAX]; };
int score_launch_gemm_bf16x3(int trans, int wm, const GemmGroup& g, const float* bias, int flags, float keep,
                             const uint8_t* mask, uint64_t seed, hipStream_t s);
// deferred weight-gradient products C = A^T . B (layout 2) of a backward pass: queued while the pass runs
// (their operands must stay untouched until the flush), then issued as one grouped launch per kernel family
// plus one launch that reduces every split-K slab.
struct GemmQueueJob { const float* A; const float* B; float* C; int M, N, K, lda, ldb, ldc; };
struct GemmQueue { int n; GemmQueueJob j[2 * GEMM_GROUP_MAX]; };
int gemm_queue_add(GemmQueue* q, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                   int ldc);
// every split-K slab set of a flushed queue: C = sum_z slab[z]  (slab order); blocks = workgroups of 256 its reduce needs
struct ReduceJob { const float* slab; float* C; int ns, M, N, ldc, first_block, pad; };
struct ReduceGroup { int n; int blocks; ReduceJob j[2 * GEMM_GROUP_MAX]; };
// defer == nullptr: the slab reduce follows the products on `s`.  defer given: the products only; *defer describes the reduce
// the caller still owes (score_launch_finish, behind the products)
// with_colsums (+ cs_part): the queued column sums' FIRST stage runs in the same launch as the f32 products (*colsums_done = 1 if
// it did: score_launch_finish(..., stage1_done) then starts at the second stage)
struct ColsumJobs;
int gemm_queue_flush(GemmQueue* q, int x3, float* slab, int64_t slab_floats, hipStream_t s, ReduceGroup* defer = nullptr,
                     const ColsumJobs* with_colsums = nullptr, float* cs_part = nullptr, int64_t cs_part_floats = 0,
                     int* colsums_done = nullptr);
#define COLSUM_MAX_JOBS 24
#define COLSUM_MAX_PARTS 128
struct ColsumJob { const float* X; float* out; const float* scale; int M, N, ld, acc, cols, rpb, nparts; int64_t part_off; };
struct ColsumJobs { ColsumJob job[COLSUM_MAX_JOBS]; int n; int64_t part_used; };
// deferred column sums: queue jobs during a pass, run them all in two launches at its end.
// The queued X matrices must stay untouched until the flush.
// scale (optional, [M]): out[n] = sum_m scale[m] * X[m][n] -- a product X^T s with one output column, without a GEMM launch
int colsum_queue_add(ColsumJobs* q, const float* X, int M, int N, int ld, float* out, int acc, const float* scale = nullptr);
int colsum_queue_flush(ColsumJobs* q, float* part, int64_t part_floats, hipStream_t s);
// the end of a backward pass in TWO launches instead of four: the queued column sums' first stage, then ONE launch that is
// the deferred split-K slab reduce of gemm_queue_flush (rg, may be empty) AND the column sums' second stage -- the two are
// independent of each other, each workgroup does what its index says.  Same arithmetic, same order as the separate launches.
// w1: the folded first attention layer's gradient computed by the same launch from the dweff / dwq products' slabs
// (instead of score_launch_attn_w1_grad behind it); stage1_done: see gemm_queue_flush
int colsum_queue_stage1(const ColsumJobs* q, float* part, int64_t part_floats, hipStream_t s);
struct W1Fold { int Dk, NA; const float* dweff; const float* dwq; float* gW1; const float* slab_e; const float* slab_q; int ns_e, ns_q; };
int score_launch_finish(const ReduceGroup* rg, ColsumJobs* q, float* part, int64_t part_floats, hipStream_t s, int stage1_done = 0,
                        const W1Fold* w1 = nullptr);
int score_launch_colsum(const float* X, int M, int N, int ld, float* out, int accumulate,
                        float* scratch, int64_t scratch_floats, hipStream_t s);
// embed.hip
struct CoattnCall {
  const int32_t* idx1; const int32_t* idx2;
  const float* tgt; int ldt;
  const int32_t* tidx;                        // forward, optional: the target's ids [B, F] -- the target rows are then read from
                                              // the table (the same bits as a gathered copy in tgt, which is not read)
  const float* W; const float* bias;
  float* out1; int ld1; float* out2; int ld2;
  float* info; int ldi; float* rsave;
  const float* g1; const float* g2; const float* ginfo;
  float* dzsum; float* slab;
  float* pcoef; float* dzcoef;                // backward, pull mode: softmax p_i and dz_i per (unit, i)
  int F, GS, nslots, first_block;
  int bit1, bit2;                             // bits of *CoattnArgs.id_status that name idx1 / idx2 (position in the feed tuple)
};
struct CoattnArgs {
  CoattnCall c[2];
  const float* table; float* gtable;
  int D4, K, T, n_units, mode;
  int Tidx;                                   // time stride of the index tensors ([B, Tidx, K, F]); 0 = T.  T = slices computed
  // ids outside [0, n_rows) (tf.nn.embedding_lookup raises there, score.py:51-66) are read as the dummy row 0 and
  // reported: bit1 / bit2 of the call OR-ed into *id_status (optional device word, score_state_t.id_status)
  uint32_t n_rows; int32_t* id_status;
};
int score_coattn_fwd_multi(CoattnArgs& a, int ncalls, int D, int B, hipStream_t s);
int score_coattn_bwd_multi(CoattnArgs& a, int ncalls, int D, int B, float* const dW[2], float* scratch,
                           int64_t scratch_floats, int atomic_scatter, ColsumJobs* cq, hipStream_t s);
int score_launch_target_fwd(const float* table, int D, int Fu, int Fi, int B, const int32_t* tu,
                            const int32_t* ti, float* query, int ldq, float* head, int ldh,
                            int off_ti, int off_tu, hipStream_t s, int64_t n_rows = 0, int32_t* id_status = nullptr);
int score_launch_target_bwd(float* grad_table, int D, int Fu, int Fi, int B, int T, const int32_t* tu,
                            const int32_t* ti, const float* dquery, int ldq, const float* dhead, int ldh,
                            int off_ti, int off_tu, const float* query, const float* W1, const float* W2,
                            const float* dzsum1, const float* dzsum2, float* S /*[2][B]*/,
                            float* dW1, float* dB1, float* dW2, float* dB2, float* dtgt_out, float* scratch,
                            int64_t scratch_floats, ColsumJobs* cq, struct GemmQueue* gq, hipStream_t s,
                            int64_t n_rows = 0);
// head.hip
int score_launch_attn_build_inp(int B, int T, int H, int NI, const float* q, const float* ur, const float* ir,
                                const float* info, float* inp, hipStream_t s);
int score_launch_attn_pool_fwd(int B, int T, int H, int NA, const float* a2, const float* w5, const float* b5,
                               const int32_t* length, const float* ur, const float* ir, float* score, float* head,
                               int ldh, int off_u, int off_i, hipStream_t s);
int score_launch_attn_tail_fwd(int B, int T, int H, int N1, int N2, const float* a1, const float* W4, const float* b4,
                               const float* w5, const float* b5, const int32_t* length, const float* ur, const float* ir,
                               float* a2, float* score, float* head, int ldh, int off_u, int off_i, hipStream_t s);
int score_launch_attn_pool_bwd(int B, int T, int H, int NA, const float* a2, const float* w5,
                               const int32_t* length, const float* ur, const float* ir, const float* score,
                               const float* dhead, int ldh, int off_u, int off_i, float* ds, float* da2,
                               hipStream_t s, int N1 = 0, const float* W4 = nullptr, const float* a1 = nullptr,
                               float* da1 = nullptr);
int score_launch_attn_inp_bwd(int B, int T, int H, int NI, const float* dinp, const float* q, const float* ur,
                              const float* ir, const float* info, const float* score, const float* dhead, int ldh,
                              int off_u, int off_i, const float* dqd, float* dur, float* dir, float* dinfo, float* dq,
                              hipStream_t s);
int score_launch_attn_fold_w1(int Dk, int NA, const float* W1, float* weff, float* wq, hipStream_t s, int copies = 1,
                              int64_t copy_stride = 0);
int score_launch_attn_inp_bwd_fused(int B, int T, int H, int NI, int N1, const float* da1, const float* Weff, const float* q,
                                    const float* ur, const float* ir, const float* info, const float* score,
                                    const float* dhead, int ldh, int off_u, int off_i, float* dur, float* dir, float* dinfo,
                                    float* dq, hipStream_t s, int N2 = 0, const float* a2 = nullptr, const float* a1 = nullptr,
                                    const float* w5 = nullptr, const float* W4 = nullptr, const int32_t* length = nullptr,
                                    float* ds = nullptr, float* da2 = nullptr, float* da1_out = nullptr);
bool score_attn_inp_bwd_fused_fits(int B, int T, int H, int NI, int N1, int N2, int ldh, int off_u, int off_i, bool pool);
#define SCORE_WEFF_COPIES 8       /* replicas of the folded attention weight (head.hip: attn_fold_w1_kernel) */
int score_launch_attn_dzsum(int B, int T, int NA, const float* dz, float* dzsum, hipStream_t s);
int score_launch_attn_w1_grad(int Dk, int NA, const float* dweff, const float* dwq, float* gW1, hipStream_t s);
int score_launch_bn_fwd(int B, int Dh, const float* x, const float* gamma, const float* beta, float rs, float* y,
                        hipStream_t s);
int score_launch_bn_bwd(int B, int Dh, const float* x, const float* gamma, float rs, const float* dy, float* dx,
                        float* dgamma, float* dbeta, float* tmp, float* scratch, int64_t scratch_floats,
                        ColsumJobs* cq, hipStream_t s);
int score_launch_l2_partials(const float* wreg, int64_t n_reg, float* part /* 256 floats */, hipStream_t s);
// the per-step transforms of the weights in one launch: [Wx_gates | Wx_cand] copies, the folded first attention layer
// (W1 = null: none), the L2 partial sums
int score_launch_weight_prep(const float* gk0, const float* ck0, const float* gb0, const float* cb0, const float* gk1,
                             const float* ck1, const float* gb1, const float* cb1, int I0, int I1, int Imax, int H, float* cat,
                             int Dk, int NA, const float* W1, float* weff, float* wq, int copies, int64_t copy_stride,
                             const float* wreg, int64_t n_reg, float* part, hipStream_t s);
int score_launch_head_out(int B, int NF, const float* f2, const float* w3, const float* b3, const int32_t* label,
                          float* logit, float* y, float* lossb, float* dlogit, float* loss, float lambda,
                          const float* part, int Bglobal, hipStream_t s, const int32_t* id_status = nullptr);
int score_launch_loss_final(int B, const float* lossb, float* loss, float lambda, const float* part, int Bglobal,
                            hipStream_t s, const int32_t* id_status = nullptr);
// head_fused.hip: bn1 + fc1 + fc2 + fc3 + sigmoid + loss terms in one launch (SCORE_E_SHAPE: shape not covered)
int score_launch_head_fwd_fused(int B, int Dh, int N1, int N2, const float* x, const float* gamma, const float* beta,
                                float rs, const float* W1, const float* b1, const float* W2, const float* b2,
                                const float* W3, const float* b3, float keep, const uint8_t* mask0, const uint8_t* mask1,
                                uint64_t seed0, uint64_t seed1, const int32_t* label, float* bn, float* f1, float* f2,
                                float* logit, float* y, float* lossb, float* dlogit, int Bglobal, hipStream_t s,
                                const uint64_t* seed_dev = nullptr, float* dz2 = nullptr, int single_launch = 0);
bool score_head_fwd_fused_fits(int B, int Dh, int N1, int N2);
int score_launch_head_bwd_fused(int B, int Dh, int N1, int N2, const float* dz2, const float* W2, const float* f1, float keep,
                                const float* W1, const float* x, const float* gamma, float rs, float* dz1, float* dbn,
                                float* dhead, float* tmp, hipStream_t s);
// head_fused.hip: attention input rows + dense_3 (folded) + dense_4 + dense_5 + masked softmax + pooling in one launch
int score_launch_attn_fwd_fused(int B, int T, int H, int NI, int N1, int N2, const float* q, const float* ur,
                                const float* ir, const float* info, const float* Weff, const float* qz, const float* W4,
                                const float* b4, const float* w5, const float* b5, const int32_t* length, float* inp,
                                float* a1, float* a2, float* score, float* head, int ldh, int off_u, int off_i,
                                hipStream_t s, int weff_copies = 1, int64_t weff_copy_stride = 0);
int score_launch_outer_relu_bwd(int B, int NF, const float* dlogit, const float* w, const float* f, float keep,
                                float* dz, hipStream_t s);
int score_launch_gru_wxcat(const float* gk0, const float* ck0, const float* gb0, const float* cb0, const float* gk1,
                           const float* ck1, const float* gb1, const float* cb1, int I0, int I1, int Imax, int H,
                           float* cat, hipStream_t s);
int score_launch_copy2d(int64_t rows, int cols, const float* src, int lds_, float* dst, int ldd, hipStream_t s);

// scatter.hip: occurrence sort ("index plan") and the pull-form gradient scatter
struct PlanFillArgs {
  const int32_t* idx[6];   // user_1hop, item_2hop, user_2hop, item_1hop, target_user, target_item
  int64_t off[7];          // prefix offsets of the six tensors in the occurrence space
  int F[6];
  int K, G, shift;         // G > 1: key = (row % G) << shift | row / G
  int T, TA;               // index tensors are [B, T, K, F]; only slices t < TA are enumerated (bt = b * TA + t)
  uint32_t n_rows;         // ids >= n_rows (feature_size) become the dummy row 0 and are reported in *id_status
  int32_t* id_status;      // optional device word (score_state_t.id_status)
};
struct PullArgs {
  const float* G[6]; int ldg[6]; int gcol[6];     // activation-gradient matrix per segment
  const float* cA[6]; const float* cB[6];         // per-(unit,k) scalars (null: constA / none)
  const float* Wv[6];                             // co-attention weight slice multiplied by cB
  float constA[6];
  const uint32_t* uid;     // null: a run's destination is its row id; else the run's unique position
  int D, K, LPRp;
  int zero_is_dummy;       // key 0 is the masked dummy row (score.py:44-47): no gradient
  uint8_t* flags;          // optional per-destination-row state byte: 2 = written this step
};
struct PlanRemapArgs { int32_t* out[6]; int F[6]; int K; int T, TA; };
int score_launch_plan_unique(const PlanRemapArgs& ra, const uint32_t* keys, const uint32_t* vals, int64_t n,
                             uint32_t* flags_scratch, uint32_t* uid, uint32_t* unique_keys, int32_t* unique_rows,
                             int32_t* meta, int G, int shift, void* temp, size_t temp_bytes, hipStream_t s,
                             bool remap = true);
int score_scan_temp_bytes(int64_t n, size_t* bytes);
int score_plan_temp_bytes(int64_t n, int end_bit, size_t* bytes);
int score_launch_plan(const PlanFillArgs& a, int key_bits, uint32_t* keys_in, uint32_t* vals_in, uint32_t* keys_out,
                      uint32_t* vals_out, void* temp, size_t temp_bytes, hipStream_t s, int which = 0);
// sort.hip: the stable radix sort of the plan (small batches) and its scratch
int score_launch_plan_own(const PlanFillArgs& a, int key_bits, uint32_t* keys_in, uint32_t* vals_in, uint32_t* keys_out,
                          uint32_t* vals_out, void* temp, size_t temp_bytes, hipStream_t s);
size_t score_sort_temp_bytes(int64_t n);
int score_pull_window(int64_t n);
int score_launch_pull(PullArgs& a, const uint32_t* keys, const uint32_t* vals, int64_t n, float* out,
                      float* partials, int64_t partial_floats, hipStream_t s);

// gru.hip: both GRUs of the model in one launch
struct GruSide {
  const float* xproj;            // [B*T, 3H]  hoisted x.Wx + b
  const float* Wg; int ldwg;     // h-rows of gates/kernel     [H, 2H]
  const float* Wc; int ldwc;     // h-rows of candidate/kernel [H, H]
  float* out; int ldo;           // [B*T, H] h_t (0 past the length)
  float* gates;                  // [B*T, 3H] saved (r, u, c)
  float* final_state;            // [B, H] or null
  // backward
  const float* dout; int lddo;   // dL/d out
  const float* dfinal;           // dL/d final state or null
  float* dxproj;                 // [B*T, 3H] pre-activation grads
  float* rh;                     // [B*T, H] r * h_{t-1}
  float* hprev;                  // [B*T, H] h_{t-1}  (+ 3*H*H scratch floats at its end for the fallback)
  float* bias_slab;              // optional [workgroups of this side][3H]: column sums of the dxproj rows a workgroup wrote
                                 // (the bias gradients' partial sums; only kernels that report GruArgs.bias_slab_rows fill it)
};
struct GruArgs {
  GruSide s[2];
  const int32_t* length;
  int B, T, H;
  int nw8;      // H = 128: 8 waves per workgroup (2 per SIMD) instead of 4
  // hidden sizes without a register-resident kernel: H = 256 streams its weights from L2 (gru_stream.hip; scratch
  // = score_gru_stream_tmp_floats); others run the recurrence step by step (two grouped GEMMs + two pointwise
  // launches per time slice; scratch 10 * B * H floats), x3 = bf16x3 allowed there.  stepwise != 0 forces that path (A/B)
  float* tmp; int64_t tmp_floats; int x3; int stepwise;
  int x3_rec;   // H = 128: the recurrence itself on the bf16 matrix cores, fp32-accurate (gru_x3.hip)
  int bias_slab_rows;   // out (backward): rows of GruSide.bias_slab written per side, 0 if the kernel that ran does not
};
// nprob same-shape GEMMs C_i = op(A_i) op(B_i) in one launch (the two sides of a recurrence step); flags: 4 = C += .
int score_gemm_same_shape(int trans, int nprob, int M, int N, int K, const float* const* A, int lda,
                          const float* const* B, int ldb, float* const* C, int ldc, int flags, int x3, float* scratch,
                          int64_t scratch_floats, hipStream_t s, const float* const* bias = nullptr);   // flags: 1 = + bias[i][N]
// gemm_panel.hip: bf16x3 products C = A . Bt^T (+ bias) with a whole-N output panel per workgroup (the GRU input projections
// and their input gradients at cfg-3's sizes): A is read once, the weights come as fragment images written once per step
struct PanelGroup { const float* A; const float* img; float* C; const float* bias; };
bool score_gemm_panel_ok(int ngroups, int M, int N, int K, int lda, int ldc, int* mt_out);      // false: use the tiled kernels
int64_t score_gemm_panel_image_floats(int N, int K);                                          // 0: shape not covered
// images of nimg weight matrices of one shape; Bt(n, k) = trans ? B[k * ldb + n] : B[n * ldb + k]
int score_gemm_panel_prep(int nimg, const float* const* B, int ldb, int trans, int N, int K, float* const* img, hipStream_t s);
int score_gemm_panel(int ngroups, const PanelGroup* g, int M, int N, int K, int lda, int ldc, hipStream_t s);
// gru_stream.hip: H = 256 (weights streamed from L2 in MFMA fragment order; tmp holds the fragment copies)
bool score_gru_stream_ok(int H);
int64_t score_gru_stream_tmp_floats(int H, int nsides);
int score_gru_fwd_stream(GruArgs& a, int nsides, hipStream_t s);
int score_gru_bwd_stream(GruArgs& a, int nsides, hipStream_t s);
// gru_x3.hip: H = 128 recurrences as bf16x3 products (weights resident in registers as split planes)
bool score_gru_x3_ok(int H, int nw8);
int score_gru_fwd_x3(GruArgs& a, int nsides, hipStream_t s);
int score_gru_bwd_x3(GruArgs& a, int nsides, hipStream_t s);
int score_gru_fwd_multi(GruArgs& a, int nsides, hipStream_t s);
int score_gru_bwd_multi(GruArgs& a, int nsides, hipStream_t s);
