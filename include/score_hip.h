/*
 * score_hip.h -- C-ABI of libscore_hip.so: the MI355X (gfx950) implementation of
 * the SCoRe forward/backward hot path (reference: qinjr/SCoRe code/score/score.py).
 *
 * The reference is pure Python on TensorFlow 1.x and has no FFI of its own
 * ("Native components: NONE"); what a reference maintainer
 * binds instead of `sess.run(...)` is this header (INTEGRATION.md shows the
 * ctypes stub).  Each entry point cites the reference lines it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch tensors in
 *     this repo); the library never allocates, frees or retains device memory;
 *   - sizes are int64_t / int32_t, `stream` is a hipStream_t passed as void*;
 *   - return value: 0 = ok, >0 = hipError_t, <0 = SCORE_E_* argument error;
 *   - state between calls: none, except the side stream + three events that score_forward/backward fork
 *     independent work onto.  They live in a score_context_t the caller creates, passes in
 *     score_state_t.context and destroys; calls with DIFFERENT contexts are re-entrant across streams and
 *     host threads.  A NULL context selects one process-wide default context per device (created on
 *     first use under a mutex, released by score_context_destroy(NULL)): calls sharing it must not
 *     overlap in time.  The library reads no environment variable (A/B switches: score_state_t.debug_flags);
 *   - all arithmetic is fp32, all indices int32 (score.py:21-30 placeholders);
 *   - table row 0 is the dummy node and must be all-zero in `table`
 *     (score.py:44-47 emb_mtx * mask): kernels rely on it and never update it.
 */
#ifndef SCORE_HIP_H
#define SCORE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCORE_E_BADARG   (-1)  /* null pointer / non-positive size               */
#define SCORE_E_SHAPE    (-2)  /* shape outside what the kernels are built for   */
#define SCORE_E_WORKSPACE (-3) /* workspace too small                            */
#define SCORE_E_INDEX    (-4)  /* a feature id outside [0, feature_size) was fed (score_id_status) */

/* Per-caller resources of the whole-path entry points (one non-blocking HIP stream + three events on the
 * current device).  Replaces nothing in the reference: it is what `tf.Session` owns there (train_score.py:188). */
typedef void* score_context_t;
int score_context_create(score_context_t* ctx);
/* ctx == NULL: release the process-wide default contexts.  Synchronises the context's stream first. */
int score_context_destroy(score_context_t ctx);
/* Events for stream-to-stream ordering ONLY (round 6): hipEventDisableTiming | hipEventDisableSystemFence -- recording one does
 * not write back / invalidate the caches at system scope, which an event made for the host's eyes does on every record (a dozen
 * records per training step sit between dependent kernels of the launch stream).  What they order is device work on device
 * memory; a host thread that wants to READ results behind such an event must use an ordinary event (or synchronise the stream).
 * The context's own fork / join events are of this kind.  score_event_record / score_stream_wait_event take any hipEvent_t.
 * Replaces nothing in the reference. */
int score_event_create(void** event);
int score_event_destroy(void* event);
int score_event_record(void* event, void* stream);
int score_stream_wait_event(void* stream, void* event);
int score_event_query(void* event);          /* 0 = reached, 1 = not yet, else an error */
int score_event_synchronize(void* event);

/* The scalars of a training step that change from step to step, kept in DEVICE memory so that a captured step
 * (hipGraph: small shapes are launch-bound, ~60 launches of a few microseconds each) can be replayed with new
 * values: ApplyAdam's alpha = lr * sqrt(1 - beta2^t) / (1 - beta1^t) (score.py:96-99) and the seed of the
 * step's dropout masks (score.py:71,73).  The caller rewrites them (one 16-byte copy) before every step. */
typedef struct {
  float    adam_alpha;
  uint32_t reserved;
  uint64_t drop_seed;
} score_step_scalars_t;

/* model_type values (train_score.py:170-179 selects the class by name) */
enum { SCORE_MODEL_SCORE = 0, SCORE_MODEL_RIA = 1, SCORE_MODEL_RCA = 2,
       SCORE_MODEL_SCORE_USER = 3, SCORE_MODEL_SCORE_ITEM = 4,
       SCORE_MODEL_RRN = 5 /* slice_models/slice_model.py:155-174: summed 1-hop sets -> GRUs -> head */ };

/* Constructor arguments of SCOREBASE.__init__ (score.py:12-13). */
typedef struct {
  int64_t feature_size;        /* N: rows of emb_mtx                              */
  int32_t eb_dim;              /* D: multiple of 4, <= 256                        */
  int32_t hidden_size;         /* H                                               */
  int32_t max_time_len;        /* T                                               */
  int32_t obj_per_time_slice;  /* K <= 32                                         */
  int32_t user_fnum;           /* Fu                                              */
  int32_t item_fnum;           /* Fi                                              */
  int32_t model_type;          /* SCORE_MODEL_*                                   */
} score_config_t;

/* One dense variable inside the flat parameter buffer (TF variable names,
 * score.py graph-construction order). */
typedef struct {
  char    name[64];
  int64_t offset;              /* in floats, into the flat buffer                 */
  int32_t rows, cols;          /* cols == 0 for vectors                           */
  int32_t regularised;         /* build_l2norm name filter, score.py:91-94        */
  int32_t init;                /* 0 zeros, 1 ones, 2 glorot-uniform               */
} score_param_entry_t;

/* Layout of the flat dense-parameter buffer: regularised tensors first
 * ([0, n_reg) floats), then biases.  Returns the number of entries (or <0). */
int score_param_layout(const score_config_t* cfg, score_param_entry_t* out, int32_t max_entries,
                       int64_t* n_floats, int64_t* n_reg_floats);

/* The batch handed to SCOREBASE.train/eval (score.py:101-133, graph_loader.py:383),
 * already on the device as row-major int32. */
typedef struct {
  const int32_t* user_1hop;    /* [B,T,K,Fi] */
  const int32_t* user_2hop;    /* [B,T,K,Fu] */
  const int32_t* item_1hop;    /* [B,T,K,Fu] */
  const int32_t* item_2hop;    /* [B,T,K,Fi] */
  const int32_t* target_user;  /* [B,Fu]     */
  const int32_t* target_item;  /* [B,Fi]     */
  const int32_t* label;        /* [B]        */
  const int32_t* length;       /* [B]        */
  int32_t B;
  int32_t active_slices;       /* 0 (or >= T) = all T time slices.  0 < A < T: the caller promises
                                  length[b] <= A for every sample, and only slices t < A are gathered,
                                  planned and computed; every [B*T, .] activation of the pass is then
                                  laid out [B*A, .].  Slices t >= length[b] reach nothing in the
                                  reference either: dynamic_rnn(sequence_length) zeroes their outputs
                                  (score.py:205-208) and the temporal attention masks them to an exact 0
                                  weight (:182-185), so loss, predictions and every gradient are
                                  unchanged; samples with length[b] > A are treated as length A.
                                  The index tensors keep their [B,T,K,F] strides.                   */
} score_batch_t;

/* Named float offsets into the workspace (for tests / introspection). */
typedef struct {
  int64_t total_bytes;
  int64_t xside;      /* [2][B*T][Di+Du]  user_side, item_side (score.py:200-201) */
  int64_t atten_info; /* [B*T][4K]        [info_item | info_user] (score.py:198)          */
  int64_t rsave;      /* [2][B*T][K]      relu'd relateness r_i per co-attention  */
  int64_t query;      /* [B][Du+Di]       [target_user, target_item] (:210)       */
  int64_t head_inp;   /* [B][Dhead]       (:217)                                  */
  int64_t att_score;  /* [B][T]           (:213)                                  */
  int64_t logit;      /* [B]                                                      */
  int64_t y_pred;     /* [B]              (:76)                                   */
  int64_t loss;       /* [4]: loss, log_loss, l2, unused                          */
  int64_t gru_out;    /* [2][B*T][H]      user_side_rep_t, item_side_rep_t        */
  int64_t gru_final;  /* [2][B][H]        final states (RIA, score.py:244-247)    */
  /* index plan (score_index_plan) */
  int64_t n_occurrences;    /* R*B row uses in the batch (SURVEY 8d's R per sample)          */
  int64_t plan_meta;        /* int32 [2+G]: [0] = U unique rows (incl. row 0), [1+o] = first
                               unique position owned by shard o (o = 0..G)                   */
  int64_t plan_unique_rows; /* int32 [U]: local row index (row / G) of every unique row,
                               grouped by owner shard (row % G), ascending                   */
  int64_t plan_remap[6];    /* int32 copies of user_1hop, item_2hop, user_2hop, item_1hop,
                               target_user, target_item holding unique positions            */
} score_workspace_t;

int score_workspace_layout(const score_config_t* cfg, int32_t B, score_workspace_t* out);
/* Float offset of an INTERNAL workspace region by name ("gates", "a1", "dxproj", "dtgt", ...: csrc/engine.hip lists them) for
 * tests and tools that compare two forms of a pass region by region; *second (optional) gets the offset of the second
 * side / call of a paired region, -1 if the region is single.  Host-only; SCORE_E_BADARG for an unknown name.  Replaces
 * nothing in the reference (what `sess.run([tensor])` on an intermediate would be, score.py:188-224). */
int score_workspace_field(const score_config_t* cfg, int32_t B, const char* name, int64_t* offset, int64_t* second);

/* ---- per-op entry points (each is also a stage of score_forward/backward) ---- */

/* Measurement helper (bench.py's `roofline.peak_measured`, SURVEY.md 8d "the measured stream-copy ceiling as a second
 * denominator"): dst[i] = src[i] for n_floats floats (a multiple of 4; both 16-byte aligned) as a plain float4 grid-stride
 * copy -- the HBM bandwidth a trivially coalesced kernel reaches on this box (n_floats * 8 bytes move per call). */
int score_stream_copy(float* dst, const float* src, int64_t n_floats, void* stream);

/* tf.truncated_normal_initializer (defaults: mean 0, stddev 1, resampled outside 2 sigma; score.py:44) for a
 * row shard of emb_mtx: local row i holds global row i * row_stride + row_first (row_stride = number of shards,
 * row_first = this shard's rank; 1 / 0 for the whole table).  Every element is a pure function of
 * (seed, global row, column) -- the inverse-CDF of a counter-based uniform -- so any sharding of the same seed
 * yields the same table, and no rank ever materialises rows it does not own.  Global row 0 (the masked dummy
 * row, score.py:45-47) and local rows past n_global_rows are written as zeros. */
int score_table_init(float* table, int64_t n_local_rows, int32_t D, int64_t row_stride, int64_t row_first,
                     int64_t n_global_rows, uint64_t seed, void* stream);

/* tf.nn.embedding_lookup + reshape (score.py:51-66): out[r, :] = table[idx[r], :].
 * Bit-exact copy. n_idx rows of D floats. */
int score_gather_fwd(const float* table, int64_t n_rows, int32_t D, const int32_t* idx,
                     int64_t n_idx, float* out, void* stream);

/* Fused gather + co_attention (score.py:147-167 in its exact collapsed form,
 * SURVEY.md 8a A4) for ONE call: seq1/seq2 index tensors [BT,K,F], target rows
 * gathered to tgt [B, F*D].  Writes seq1_result to out1 (row stride ld1),
 * seq2_result to out2 (ld2), atten_info [BT,2K] into info (ldi) and r [BT,K]
 * into rsave.  W = dense kernel [3*F*D], bias = [1].  mode 0 = co-attention,
 * mode 1 = RCA's reduce_sum over K (score.py:266-269; W/bias/info/rsave unused). */
int score_coattn_fwd(const float* table, int64_t n_rows, int32_t D, int32_t F, int32_t K,
                     int32_t B, int32_t T, const int32_t* idx1, const int32_t* idx2,
                     const float* tgt, const float* W, const float* bias,
                     float* out1, int32_t ld1, float* out2, int32_t ld2,
                     float* info, int32_t ldi, float* rsave, int32_t mode, void* stream);

/* Backward of the above.  g1/g2/ginfo are the gradients of out1/out2/info with
 * the same strides.  Scatter-adds row gradients into grad_table [n_rows, D]
 * (row 0 skipped: mask, score.py:47), writes dzsum [BT] (sum_i dz_i, feeds the
 * target-row / w_t / bias gradients) and accumulates dW[Dx..3Dx) into
 * partial slabs reduced by the library into dW (+=). */
int score_coattn_bwd(const float* table, float* grad_table, int64_t n_rows, int32_t D, int32_t F,
                     int32_t K, int32_t B, int32_t T, const int32_t* idx1, const int32_t* idx2,
                     const float* W, const float* rsave,
                     const float* g1, int32_t ld1, const float* g2, int32_t ld2,
                     const float* ginfo, int32_t ldi, float* dzsum, float* dW,
                     float* scratch, int64_t scratch_floats, int32_t mode, void* stream);

/* C[M,N] = epi(op(A) . op(B)) in exact fp32 (v_mfma_f32_32x32x2_f32).
 * trans: 0 = A[M,K] B[K,N];  1 = A[M,K] B[N,K]^T;  2 = A[K,M]^T B[K,N] (K is the
 * reduced dim).  flags: 1 add bias[N], 2 relu, 4 accumulate into C,
 * 8 dropout (tf.nn.dropout: x/keep * mask; mask from drop_mask bytes or, if null,
 * from a counter hash of drop_seed), 16 use the bf16x3 matrix-core product (fp32-accurate,
 * see score_state_t.gemm_mode) on the shapes where it measured faster, 32 use it whenever legal,
 * 64 relu backward fused: drop_mask then points at the layer's fp32 output Y [M,N] and the result is
 * C = [Y > 0] * (A.B) / keep_prob (not together with 8).
 * Bits 16-30: bias row group g (0 = the usual single bias row): bias is [ceil(M/g), N] and
 * output row r adds bias row r/g -- the per-sample term of the folded attention layer.
 * scratch is used for split-K. */
int score_gemm(int32_t trans, int32_t M, int32_t N, int32_t K,
               const float* A, int32_t lda, const float* Bm, int32_t ldb,
               float* C, int32_t ldc, const float* bias, int32_t flags,
               float keep_prob, const uint8_t* drop_mask, uint64_t drop_seed,
               float* scratch, int64_t scratch_floats, void* stream);

/* The same-shape products of `ngroups` (<= 4) independent problems C_i[M,N] = A_i[M,K] . op(B_i) (+ bias_i[N]) in ONE launch
 * on the bf16 matrix cores, fp32-accurate (bf16x3), with a whole-N output panel per workgroup: the form the GRU input
 * projections of the two sides (score.py:205-208) and their input gradients take at cfg-3's sizes (csrc/gemm_panel.hip).
 * trans_b: 0 = B_i[K,N], 1 = B_i[N,K] (C = A . B^T).  `images` is scratch for the weights' MFMA-fragment images,
 * ngroups * K/32 * 8*ceil(N/128) * 768 floats.  Shapes it does not cover (N % 16, N <= 256 or > 512, K % 32, ld % 4, too few
 * rows to fill the chip) return SCORE_E_SHAPE: use score_gemm. */
int score_gemm_panel_products(int32_t trans_b, int32_t ngroups, int32_t M, int32_t N, int32_t K,
                              const float* const* A, int32_t lda, const float* const* Bm, int32_t ldb,
                              float* const* C, int32_t ldc, const float* const* bias,
                              float* images, int64_t image_floats, void* stream);
/* The two halves of score_gemm_panel_products: the images of the weights (once per set of weights), and the products
 * from prepared images (as the engine runs them: the images are written once per step, beside the gather). */
int score_gemm_panel_images(int32_t trans_b, int32_t ngroups, int32_t N, int32_t K, const float* const* Bm, int32_t ldb,
                            float* images, int64_t image_floats, void* stream);
int score_gemm_panel_run(int32_t ngroups, int32_t M, int32_t N, int32_t K, const float* const* A, int32_t lda,
                         float* const* C, int32_t ldc, const float* const* bias, const float* images,
                         int64_t image_floats, void* stream);

/* tf.nn.dynamic_rnn(GRUCell(H), sequence_length) recurrence (score.py:205-208)
 * given the hoisted input projection xproj [B*T,3H] = x.[Wx_gates|Wx_cand]+bias.
 * Wg/Wc point at the h-rows of gates/kernel [H,2H] and candidate/kernel [H,H].
 * out [B*T, ldo] gets h_t (0 for t>=len), gates_save [B*T,3H] gets (r,u,c),
 * final [B,H] the carried state. */
int score_gru_fwd(int32_t B, int32_t T, int32_t H, const float* xproj, const float* Wg, int32_t ldwg,
                  const float* Wc, int32_t ldwc, const int32_t* length,
                  float* out, int32_t ldo, float* gates_save, float* final_state, void* stream);

/* Backward recurrence: dout [B*T, lddo] (+ dfinal [B,H] or null) ->
 * dxproj [B*T,3H] (pre-activation grads dr,du,dc), rh [B*T,H] = r*h_{t-1} and
 * hprev [B*T,H] = h_{t-1}; `hprev` must have room for B*T*H + 3*H*H floats (its tail
 * is scratch for the transposed recurrent weights). */
int score_gru_bwd(int32_t B, int32_t T, int32_t H, const float* Wg, int32_t ldwg, const float* Wc,
                  int32_t ldwc, const int32_t* length, const float* out, int32_t ldo,
                  const float* gates_save, const float* dout, int32_t lddo, const float* dfinal,
                  float* dxproj, float* rh, float* hprev, void* stream);

/* Guard of the optimizer entry points.  tf.nn.embedding_lookup raises inside sess.run for an id outside the table and
 * NO variable is updated by that call (score.py:51-66, 101-116).  The kernels that read the ids report such a batch in
 * the device word score_state_t.id_status; an optimizer call that is handed the same word reads it when it EXECUTES
 * (one uniform load) and, if it is non-zero, leaves the variable and both Adam slots untouched.  `skipped` (optional
 * device counter) gets +1 from each suppressed call it is passed to: the caller hands it to ONE call per step (the dense
 * variables') and learns how many steps to take off its own step count / beta powers when it sees the word.  The word
 * is sticky: every later guarded call is suppressed too until the caller clears it.  NULL guard / NULL id_status =
 * unguarded (a row shard's optimizer: the sharded path rejects such a batch on the host before the step starts). */
typedef struct {
  const int32_t* id_status;
  int32_t* skipped;
} score_guard_t;

/* tf.train.AdamOptimizer ApplyAdam (score.py:96-99), dense over n floats:
 * g' = g + l2*p (first n_reg floats); m += (g'-m)(1-b1); v += (g'^2-v)(1-b2);
 * p -= m*alpha/(sqrt(v)+eps).  alpha = lr*sqrt(1-b2^t)/(1-b1^t) from the host. */
int score_adam(float* p, float* m, float* v, const float* g, int64_t n, int64_t n_reg,
               float l2, float alpha, float beta1, float beta2, float eps, const score_guard_t* guard, void* stream);

/* score_adam / score_adam_rows with alpha read from device memory (sc->adam_alpha) at execution time. */
int score_adam_dev(float* p, float* m, float* v, const float* g, int64_t n, int64_t n_reg, float l2,
                   const score_step_scalars_t* sc, float beta1, float beta2, float eps, const score_guard_t* guard,
                   void* stream);
int score_adam_rows_dev(float* p, float* m, float* v, const float* g, int64_t n_rows, int32_t D, uint8_t* row_flags,
                        const score_step_scalars_t* sc, float beta1, float beta2, float eps, const score_guard_t* guard,
                        void* stream);

/* The same update over the [n_rows, D] embedding table, driven by a per-row state byte so the
 * dense sweep only moves the rows dense ApplyAdam actually changes (bit-identical result):
 *   0  m = v = 0 and no gradient this step  -> ApplyAdam is the identity: nothing is read
 *   1  m or v nonzero, no gradient this step -> g = 0 (moments decay, p moves); g is not read
 *   2  gradient written this step (score_backward / score_segment_sum_rows set it) -> full
 *      update from g, then the state becomes 1
 *   (3 exists only inside score_adam_catchup_ids: a live row the batch is about to read, see below)
 * g rows in state 0/1 are never read, so grad_table needs no zero fill between steps.
 * A suppressed call (guard) leaves the state bytes as they are: the caller clamps the 2s back to 1. */
int score_adam_rows(float* p, float* m, float* v, const float* g, int64_t n_rows, int32_t D,
                    uint8_t* row_flags, float alpha, float beta1, float beta2, float eps, const score_guard_t* guard,
                    void* stream);
/* score_adam_rows on the table and score_adam on the flat dense variables [n] (first n_reg regularised with l2) in ONE launch:
 * disjoint memory, the same arithmetic per element as the two calls.  guard: ONE counted call (the dense half counts a suppressed
 * step, neither half applies anything while the word is set). */
int score_adam_rows_and_dense(float* p, float* m, float* v, const float* g, int64_t n_rows, int32_t D, uint8_t* row_flags,
                              float* wp, float* wm, float* wv, const float* wg, int64_t n, int64_t n_reg, float l2, float alpha,
                              float beta1, float beta2, float eps, const score_guard_t* guard, void* stream);

/* ---- time-tiled ApplyAdam over the table ----------------------------------------------------------------------
 * Dense ApplyAdam moves every live row every step, but the update of a row WITHOUT a gradient (m *= b1, v *= b2,
 * p -= alpha_t m/(sqrt(v)+eps)) reads nothing except the row and the step's alpha.  These entry points apply such
 * updates late, in step order, the first time a row is needed: the same fp32 operations in the same order as the
 * sweep of score_adam_rows, so the table, m and v are bit-identical to it at every point where they are observed
 * -- while a row's HBM traffic drops from once per step to once per use.  Nothing is skipped: every (row, step)
 * update is executed exactly once.
 *   row_step[r]   optimizer steps already applied to row r (meaningful for rows in state 1 / 2)
 *   alpha_ring    SCORE_ADAM_RING + 1 floats: alpha of step s at [s % SCORE_ADAM_RING] (score_adam_touched writes
 *                 it), and one sticky error word at [SCORE_ADAM_RING], nonzero if a row ever lagged more steps than
 *                 the ring remembers (the caller's window sweep must keep every live row within
 *                 SCORE_ADAM_RING - 2 steps)
 * Protocol of step n (1-based), single table (no row sharding):
 *   score_adam_catchup_ids(ids of the batch, upto = n-1)      before score_forward
 *   score_adam_catchup_rows(slice n % window, upto = n-1)     any time after that call has finished and before
 *                                                            the next step's catchup_ids; may run on another
 *                                                            stream beside forward/backward/score_adam_touched
 *   score_forward, score_backward (scatter_mode 0: rows with a gradient end in state 2)
 *   score_adam_touched(step = n, alpha_n)                     state-2 rows: ApplyAdam from g, state 1, row_step n
 * and score_adam_catchup_rows(0, n_rows, upto = n) before anything else reads p, m or v. */
#define SCORE_ADAM_RING 64
typedef struct {
  float* p; float* m; float* v;   /* [n_rows, D] variable and its two Adam slots           */
  const float* g;                 /* [n_rows, D] row gradients (rows in state 2 are valid) */
  int64_t n_rows;
  int32_t D;
  int32_t reserved;
  uint8_t* row_flags;             /* [n_rows] state bytes of score_adam_rows               */
  uint32_t* row_step;             /* [n_rows]                                              */
  float* alpha_ring;              /* [SCORE_ADAM_RING + 1]                                 */
  float beta1, beta2, eps;
  int32_t reserved2;
  const int32_t* id_status;       /* optional guard word (score_guard_t.id_status).  While it is non-zero:
                                     score_adam_touched(_rows) applies nothing -- its rows go back to state 1 with
                                     row_step = step - 1 (they were current up to there), alpha_ring is not written --
                                     and the catch-up entry points replay nothing (the steps they would replay may be
                                     ones that were suppressed): the table, m and v stay as the last applied step left them */
  const int32_t* skipped_steps;   /* optional DEVICE counter (score_guard_t.skipped of the same caller): steps suppressed so
                                     far, i.e. by how much the `step` arguments run ahead of what was really applied   */
} score_adam_table_t;
int score_adam_touched(const score_adam_table_t* t, uint32_t step, float alpha, void* stream);
/* score_adam_touched(t, step, alpha) and score_adam(p, m, v, g, n, n_reg, l2, alpha, t->beta1, t->beta2, t->eps, guard) --
 * the step's whole ApplyAdam, table rows with a gradient and the flat dense variables -- in ONE launch: the two touch disjoint
 * memory and were two dependent launches at the end of every step (the reference's own batch sizes are bound by the host's
 * launch calls).  The guard is t->id_status; `skipped` (optional) counts the step when it is suppressed.  Same arithmetic in
 * the same order per element as the two calls: the same bits. */
int score_adam_touched_and_dense(const score_adam_table_t* t, uint32_t step, float alpha, float* p, float* m, float* v,
                                 const float* g, int64_t n, int64_t n_reg, float l2, int32_t* skipped, void* stream);
/* A backward pass whose gradient nobody applies: every state-2 row back to state 1.  A row that was live keeps its
 * row_step (the batch's rows were brought up to `upto` before that pass); a row that was in state 0 -- m = v = 0: its owed
 * updates are identities and any count is right for it -- gets row_step = upto, which keeps it inside the ring. */
int score_adam_unmark(const score_adam_table_t* t, uint32_t upto, void* stream);
/* score_adam_touched driven by a row LIST instead of a scan of the state bytes: rows[0 .. *n_rows_dev) (device int32,
 * the count is read on the device; max_rows bounds it for the launch) names every row that may be in state 2 -- what
 * score_index_plan(dedup = 2) leaves in the workspace (plan_unique_rows / plan_meta[0]).  Entries not in state 2 are
 * skipped (the dummy row 0); rows in state 2 that the list misses would be left for score_adam_touched. */
int score_adam_touched_rows(const score_adam_table_t* t, const int32_t* rows, const int32_t* n_rows_dev, int64_t max_rows,
                            uint32_t step, float alpha, void* stream);
/* every value of ids[0..n_ids) that names a live row (values outside [0, n_rows) are ignored, so a whole flat batch
 * buffer may be passed): replay the zero-gradient steps row_step+1 .. upto of that row */
int score_adam_catchup_ids(const score_adam_table_t* t, const int32_t* ids, int64_t n_ids, uint32_t upto, void* stream);
/* The NEXT batch's rows, one step early.  Between score_backward's row scatter of step `step` (its stage boundary 4: every
 * row that gets this step's gradient is in state 2 by then) and the next step's score_forward, on any stream: every id that
 * names a live row NOT in state 2 is replayed through `step` itself -- its gradient in this step is zero, so its update
 * needs nothing but alpha (= the alpha score_adam_touched(step) is called with; written to the ring here too).  Rows in
 * state 2 are left to score_adam_touched.  Afterwards every row the next batch reads is current up to `step`, and the
 * next step needs no score_adam_catchup_ids: its 0.07 ms (cfg-3) run beside this step's weight-gradient products instead
 * of in front of the next forward.  Must not overlap score_adam_catchup_rows on another stream (same rows). */
int score_adam_catchup_ids_through(const score_adam_table_t* t, const int32_t* ids, int64_t n_ids, uint32_t step, float alpha,
                                   void* stream);
/* the same for every state-1 row of [row_begin, row_end) */
int score_adam_catchup_rows(const score_adam_table_t* t, int64_t row_begin, int64_t row_end, uint32_t upto, void* stream);

/* ---- whole-path entry points --------------------------------------------- */

typedef struct {
  const float* table;   /* [n_table_rows, D] effective table (row 0 == 0)         */
  int64_t n_table_rows;
  const float* w;       /* flat dense parameters                                  */
  float* workspace;     /* score_workspace_layout(...).total_bytes                */
  int64_t workspace_bytes;
  int32_t scatter_mode; /* embedding-gradient scatter: 0 = sorted occurrences + per-row pull
                           (no float atomics, bitwise reproducible; needs score_index_plan
                           with 1 shard), 1 = global_atomic_add_f32 into grad_table,
                           2 = pull into the unique-row order of a sharded plan (grad_table
                           is the [U, D] gradient of the gathered mini-table)            */
  int32_t global_batch; /* samples the loss mean runs over (data parallel); 0 = batch->B  */
  int32_t gemm_mode;    /* dense layers: 0 = v_mfma_f32_32x32x2_f32 (bitwise an fp32 fmaf chain),
                           1 = "bf16x3": every fp32 operand split exactly into three bf16 and
                           six v_mfma_f32_32x32x16_bf16 per k-step -- fp32 accuracy (dropped
                           terms < 2^-24 relative) at 2.7x the matrix-core rate              */
  int32_t debug_flags;  /* A/B switches (0 in production): bit 0 = run the H = 256 recurrence step by step (grouped GEMMs +
                           pointwise launches per time slice) instead of the persistent streaming kernel; bit 1 = the
                           head's forward in one launch at every batch size; bit 2 = the H = 128 recurrences on the
                           f32-input MFMA kernels instead of the bf16x3 form; bit 3 = the GRU input projections (and
                           their input gradients) on the tiled bf16x3 kernel instead of the whole-N panel form
                           (csrc/gemm_panel.hip); bit 4 = the input gradients in the panel form at every size; bit 6 =
                           build_fc_net layer by layer instead of the fused head kernels; bit 7 = the temporal attention
                           layer by layer instead of its fused kernels (the paths shapes outside the fused kernels'
                           instantiated widths take anyway); bit 5 = the index plan sorts with the library's radix sort at
                           every size, bit 8 = with csrc/sort.hip's at every size (default: by the number of occurrences;
                           both are stable, the plan is the same bits); bit 9 (512) = never take the per-sample
                           whole-model kernels (csrc/persample.h: at H = 32, B <= 512, K <= 10, <= 48 computed slices
                           SCORE / SCORE_USER / SCORE_ITEM run each pass as ONE kernel, a workgroup per sample), bit 10
                           (1024) = take them for the forward pass only, bit 11 (2048) = for the backward pass only;
                           bit 12 (4096) = NO second stream: everything score_forward / score_backward would fork onto
                           the context's side stream runs on `stream`, in launch order (same results bit for bit; what
                           a suspected stream race is compared against -- score_amd.model inlines its own streams too);
                           bit 14 (16384) = (layer-by-layer pass) the recurrences' weight-gradient products at the END of the
                           launch stream's chain, behind the row scatter (the round-5 placement), instead of on the side stream
                           beside the co-attention backward and the scatter (round 6; same bits).  Bit 13 is unused.
                           The ONLY switches of the launch sequence: the library reads no environment variable       */
  uint8_t* row_flags;   /* optional [n_table_rows] row state of the dense table optimizer (see
                           score_adam_rows): score_backward (scatter_mode 0) marks every row it
                           writes into grad_table with 2 and leaves all other rows of grad_table
                           untouched, so grad_table needs no zero fill.  NULL = off: the caller
                           zero-fills grad_table and uses score_adam.                            */
  void* gather_done_event; /* optional hipEvent_t: score_forward records it right after the fused gather +
                           co-attention launch, so a caller can start the index plan of the same batch on
                           another stream behind the HBM-bound gather instead of beside it             */
  void* plan_done_event;   /* optional hipEvent_t (recorded by the caller after score_index_plan on its own
                           stream): score_backward waits for it just before the row scatter, the first
                           consumer of the plan -- not at its start                                   */
  const score_step_scalars_t* step_scalars; /* optional DEVICE pointer: score_forward then takes the dropout seed from it
                           (its by-value drop_seed argument is ignored; shapes the fused head kernel does not cover
                           return SCORE_E_SHAPE when keep_prob < 1), so the launch sequence holds no per-step value */
  score_context_t context; /* side stream + events of this caller (score_context_create); NULL = the
                           process-wide default context of the current device                         */
  int32_t* id_status;   /* optional DEVICE word, zero-initialised by the caller, sticky: tf.nn.embedding_lookup on the
                           CPU raises for an id outside [0, feature_size) (score.py:51-66).  Here every kernel that turns an
                           id into an address reads such an id as the dummy row 0 instead (never an out-of-bounds access,
                           with or without this word), and the kernels that see the ids as fed -- the fused gather and
                           the target-row gather of score_forward, the occurrence fill of score_index_plan -- OR bit i
                           into the word, i = position of the tensor in the feed tuple (0 user_1hop, 1 user_2hop,
                           2 item_1hop, 3 item_2hop, 4 target_user, 5 target_item; graph_loader.py:383).  No extra launch,
                           no read-back: score_forward's loss reduction reads the word and, if it is set, writes NaN
                           to loss[0] / loss[1] and the bits to loss[3], so whoever reads the loss learns of it;
                           score_id_status() is the synchronous query.  Ids of time slices >= active_slices are never
                           dereferenced and not checked.                                                         */
  void* grads_done_event; /* optional hipEvent_t (score_backward): the pass's last four launches -- slab reduce, folded attention
                           layer's gradient, column sums: everything that FINISHES grad_w -- then run on the context's side
                           stream behind the weight-gradient products, and this event is recorded behind them.  grad_table is
                           complete on `stream` when score_backward returns as before; grad_w only once the event has fired:
                           the caller runs what needs the row gradients alone (score_adam_touched) on `stream` meanwhile and
                           makes `stream` wait for the event before score_adam on the dense variables (or anything else that
                           reads grad_w).  Round 6: the finishers are forked in FRONT of the row scatter (beside it), so the
                           event says nothing about the scatter -- which reads the co-attention weights: the dense variables
                           may only be updated behind the event AND behind `stream`'s own launches of the pass (i.e. on `stream`,
                           or on a stream that also waits for stage boundary 4).  NULL: everything on `stream`, as before. */
  void* loss_done_event;  /* optional hipEvent_t (score_forward): the loss reduction (one workgroup: loss[0..3] from the per-sample
                           terms and the L2 partial sums) then runs on the context's side stream behind the head, and this event is
                           recorded behind it -- the head's successor on `stream` (score_backward's first launch) no longer queues
                           behind a one-block kernel.  loss[] is final once the event has fired; on `stream` it is final behind
                           score_backward of the same step (which joins the side stream), not behind score_forward alone.  NULL:
                           on `stream`, as before (evaluation: nothing follows the forward pass).
                           (The per-sample whole-model forward kernel reduces the loss itself -- its last workgroup to finish
                           does it, round 5 -- so there the event is simply recorded on `stream` behind that kernel.)      */
  float* loss_host;       /* optional: four floats of PINNED HOST memory mapped into the device (hipHostMalloc): the per-sample
                           forward kernel also stores loss[0..3] there (system scope), so a caller that returns the loss every
                           step (score.py:101-116) reads it after waiting for an event recorded behind score_forward -- no copy,
                           no extra stream.  Ignored by the layer-by-layer pass (score_persample_form tells which runs).   */
  float* plan_workspace;  /* optional (score_backward, scatter_mode 0): the batch's index plan lies in THIS buffer -- another
                           workspace of the same layout, where score_index_plan wrote it -- instead of in `workspace`: a caller
                           that alternates two buffers for the plans can sort the next batch's plan while this step's scatter
                           still reads its own (everything else of the step stays in `workspace`).  NULL: in `workspace`.       */
  int32_t reserved4;
  int32_t reserved3;
} score_state_t;

/* Synchronous query of a score_state_t.id_status word: copies it to *bits (optional), waits for `stream`, clears the
 * word when `clear` != 0, and returns SCORE_E_INDEX if any bit was set (0 otherwise): the status TF's
 * InvalidArgumentError "indices[...] is not in [0, N)" corresponds to (score.py:51-66). */
int score_id_status(int32_t* id_status, int32_t* bits, int32_t clear, void* stream);

/* Which kernel form the two sides' GRU input projections (score.py:205-208; *x_form) and their input gradients
 * (*dx_form) take in score_forward / score_backward for a batch of B samples with `active_slices` computed time slices
 * (0 = all T) under st->gemm_mode / st->debug_flags: 0 = the tiled kernels (gemm_bf16x3.hip / gemm.hip), n > 0 = the
 * whole-N panel form (gemm_panel.hip) with each side's output columns as n panel groups (2 = column halves: H = 256).
 * Host-only query, nothing is launched: what the parity tests assert before they compare the two forms. */
int score_gemm_forms(const score_config_t* cfg, const score_state_t* st, int32_t B, int32_t active_slices,
                     int32_t* x_form, int32_t* dx_form);

/* 1 if score_forward / score_backward run a batch of B samples with `active_slices` computed time slices (0 = all T) as the
 * per-sample whole-model kernels (csrc/persample.h) under st->scatter_mode / st->debug_flags, 0 if layer by layer.  Host-only:
 * callers use it to place their own side-stream work (score_amd/model.py), tests to assert the form they compare. */
int score_persample_form(const score_config_t* cfg, const score_state_t* st, int32_t B, int32_t active_slices);

/* Index plan of a batch (depends on the indices only; run it before score_backward, on
 * any stream ordered before it).  Radix-sorts all R*B row uses by (owner shard, row).
 * n_shards > 1 (table row-sharded, owner = row % n_shards) or dedup != 0 additionally
 * de-duplicates them: unique rows grouped by owner -> what to request from each shard --
 * and writes the six index tensors remapped to unique positions, so the same kernels run
 * on the gathered [U, D] mini-table.  dedup == 2 (one shard): only the unique row list and its count are written
 * (plan_unique_rows, plan_meta[0]; no remapped tensors) -- the rows score_adam_touched_rows updates.  Replaces nothing in the reference (it has no multi-device code);
 * it is the index-routing step BASELINE.json's north_star asks for. */
int score_index_plan(const score_config_t* cfg, const score_state_t* st, const score_batch_t* batch,
                     int32_t n_shards, int32_t dedup, void* stream);

/* The small-batch sort behind score_index_plan as an op of its own (csrc/sort.hip): a STABLE least-significant-digit radix
 * sort of n (key, value) pairs of 32-bit words by the low key_bits bits of the key (all higher bits must be zero), 11 - 12 bits
 * per pass (key_bits 21: two passes; 32: three), six launches for two passes whatever n.  keys / vals and keys_alt / vals_alt
 * are the two buffers the passes alternate between; *result_in_alt says where the sorted pairs are (1: the alt buffers -- an
 * odd number of passes --, 0: keys / vals); the other pair holds an intermediate pass.  temp: score_sort_pairs_temp_bytes(n). */
int score_sort_pairs(uint32_t* keys, uint32_t* vals, uint32_t* keys_alt, uint32_t* vals_alt, int64_t n, int32_t key_bits,
                     void* temp, int64_t temp_bytes, int32_t* result_in_alt, void* stream);
int64_t score_sort_pairs_temp_bytes(int64_t n);

/* out[rows[j], :] = sum over slots j with equal rows[j] of src[j, :] (slot order; rows never
 * named keep their value).  The shard owner uses it to combine the row gradients received
 * from every rank.  row_flags (optional, [n_out_rows]): every row written is marked 2 for
 * score_adam_rows.  scratch: score_segment_sum_scratch_bytes(n, D). */
int score_segment_sum_rows(const int32_t* rows, const float* src, int64_t n, int32_t D, int64_t n_out_rows,
                           float* out, uint8_t* row_flags, void* scratch, int64_t scratch_bytes, void* stream);
int64_t score_segment_sum_scratch_bytes(int64_t n, int32_t D);

/* The owner's usual way to combine the row gradients it receives: one call per source rank, in rank
 * order, each with that rank's rows (unique inside a call -- they come from score_index_plan's
 * de-duplication) and gradients.  The first writer of a row this step stores, later ones add:
 * out[rows[i], :] (+)= src[i, :], reproducible without a sort or an atomic.  row_flags (required):
 * rows already in state 2 are added to, every row touched ends in state 2 (score_adam_rows). */
int score_rows_accumulate(const int32_t* rows, const float* src, int64_t n, int32_t D, int64_t n_out_rows,
                          float* out, uint8_t* row_flags, void* stream);
/* All source ranks in ONE launch: rows / src hold the sources' lists back to back, source p at [offsets[p], offsets[p+1])
 * (offsets: HOST array of n_sources + 1 entries, offsets[0] = 0; n_sources <= 64), every list unique and ascending (what
 * score_index_plan's unique-row lists are).  The result is bit for bit that of n_sources score_rows_accumulate calls in
 * source order: the slot of the lowest source that names a row adds the later sources' rows in source order (binary search
 * per list) and stores once. */
int score_rows_accumulate_multi(const int32_t* rows, const float* src, const int64_t* offsets, int32_t n_sources, int32_t D,
                                int64_t n_out_rows, float* out, uint8_t* row_flags, void* stream);

/* Forward of SCORE / RIA / RCA / SCORE_USER / SCORE_ITEM (score.py:188-369) +
 * build_fc_net / build_logloss / build_l2norm (:68-94).  keep_prob 1.0 = eval
 * (score.py:129), 0.8 = train (:113).  Results land in the workspace
 * (y_pred, loss, ...). drop_mask0/1: optional explicit [B,200]/[B,80] byte masks.
 * stage_events: null, or 5 hipEvent_t handles (each may be null) recorded on `stream`
 * at the stage boundaries: [0] before the fused gather+co-attention launch, [1] after
 * it, [2] after the GRUs, [3] after the temporal attention, [4] after head + loss. */
int score_forward(const score_config_t* cfg, const score_state_t* st, const score_batch_t* batch,
                  float reg_lambda, float keep_prob, const uint8_t* drop_mask0,
                  const uint8_t* drop_mask1, uint64_t drop_seed, void* const* stage_events,
                  void* stream);

/* Backward of the same graph: grad_w [n_floats] (overwritten; WITHOUT the L2
 * term, which score_adam adds) and grad_table [n_table_rows, D] (scatter_mode 1: zeroed by
 * the caller, accumulated into; modes 0/2: the rows of the batch are overwritten, see
 * score_state_t.row_flags).  Must follow score_forward on the same workspace/batch (and, to reuse its fork of the context's side
 * stream, on the same stream and context: otherwise the pass records one event more at its start).
 * stage_events: null, or SIX handles: [0] start, [1] after the head, [2] after the temporal
 * attention, [3] after the GRUs, [4] after the co-attention/embedding scatter, [5] after the
 * weight-gradient products (all X^T dY of the pass run here, as grouped launches; with score_state_t.grads_done_event the
 * event sits behind the products, the finishers that follow them run on the side stream).
 * grad_w between the two calls: score_backward's zero fill / first writers of grad_w run on the context's side stream, forked
 * where score_forward of the same step began -- i.e. they may execute any time after score_forward was CALLED.  Nothing may
 * read or write grad_w on `stream` (or anywhere else) between score_forward and the completion of score_backward of the same
 * step; the previous step's readers (the optimizer) must be on `stream` before score_forward, or joined into it.            */
int score_backward(const score_config_t* cfg, const score_state_t* st, const score_batch_t* batch,
                   float keep_prob, float* grad_w, float* grad_table, void* const* stage_events,
                   void* stream);

/* ---- one training step in one call (csrc/step.hip) -----------------------------------------------------------------------
 * The steady-state step of model.train (score.py:101-116) for a caller that runs the time-tiled table optimizer with the
 * look-ahead (score_adam_catchup_ids_through) and sorts the next batch's index plan a step ahead: exactly
 *   wait(ev_ahead), wait(ev_sweep)                                   [stream]   if wait_ahead / wait_sweep
 *   score_forward(..., loss_host, loss_done_event = ev_loss if loss_host)                   [stream]
 *   score_backward(plan_done_event = ev_plan, grads_done_event = ev_grads, stage event 4 = ev_b4)   [stream]
 *   wait(ev_grads); score_adam_touched_and_dense(step, alpha, l2 = reg_lambda)              [stream]
 *   wait(ev_b4); score_adam_catchup_ids_through(next ids, step, alpha); record(ev_ahead);
 *                score_index_plan(next batch -> next_workspace); record(ev_plan)            [side_stream]   if next_batch
 *                score_adam_catchup_rows(slice); record(ev_sweep)                           [side_stream]   if slice_hi > slice_lo
 *   (in this order of CALLS; on the device the side stream's work runs beside the optimizer launch)
 * -- the same entry points with the same arguments a caller would use one by one (the events are the caller's, re-used from
 * call to call: every wait above is issued before the same call re-records its event).  PRECONDITIONS the caller guarantees:
 * the batch's rows are up to date through step - 1 (the previous call named this batch as next_batch, or the caller ran
 * score_adam_catchup_ids), ev_plan is recorded behind THIS batch's index plan in st->workspace, no row is in state 2.
 * What it saves is host time between the calls (the reference's own batch sizes are bound by it), nothing on the device. */
typedef struct {
  const score_adam_table_t* table;                       /* the embedding table's optimizer state (g = grad_table of the passes) */
  float* w; float* w_m; float* w_v; float* w_g;          /* flat dense variables, their Adam slots and gradient [n_w]           */
  int64_t n_w, n_reg;
  int32_t* skipped;                                      /* optional score_guard_t.skipped                                      */
  float reg_lambda, keep_prob, alpha;
  uint32_t step;                                         /* the optimizer step being applied (1-based)                          */
  uint64_t drop_seed;
  int64_t slice_lo, slice_hi; uint32_t slice_upto;       /* the window slice of this step (score_adam_catchup_rows)             */
  int32_t wait_ahead, wait_sweep;
  const score_batch_t* next_batch; const int32_t* next_ids; int64_t n_next_ids;      /* optional: the batch the NEXT call trains on */
  float* next_workspace; int64_t next_workspace_bytes;
  float* loss_host;                                      /* optional, see score_state_t.loss_host                               */
  void* side_stream;
  void* ev_ahead; void* ev_sweep; void* ev_plan; void* ev_stage2 /* unused */; void* ev_b4; void* ev_grads; void* ev_loss;   /* hipEvent_t */
  void* ev_plan_next;     /* optional hipEvent_t.  Given (and next_workspace a DIFFERENT workspace than st->workspace: the caller
                             alternates two): the next batch's index plan is queued FIRST on the side stream, with no wait, and
                             this event recorded behind it -- the sort runs beside this step's passes.  NULL: one workspace, the
                             plan behind this step's scatter, ev_plan re-recorded (the sequence above).                         */
  void* const* fwd_stage_events;   /* optional: score_forward's stage_events (five hipEvent_t or NULL each) -- a caller timing the pass */
  void* plan_stream;      /* optional: a third stream for that early plan (it waits for ev_b4's previous record first); NULL: the
                             side stream -- the look-ahead catch-up then queues behind the sort.                                */
} score_train_step_t;
int score_train_step(const score_config_t* cfg, const score_state_t* st, const score_batch_t* batch, const score_train_step_t* p,
                     void* stream);

/* sizeof of every struct above in the header's order (score_step_scalars_t, score_config_t, score_param_entry_t, score_batch_t,
 * score_workspace_t, score_guard_t, score_adam_table_t, score_state_t, score_train_step_t, score_graph_t, score_batch_out_t), then
 * offsetof(score_state_t, plan_workspace), offsetof(score_train_step_t, plan_stream), offsetof(score_adam_table_t, skipped_steps):
 * a binding in another language checks its own structures against these before its first call.  Returns how many values it
 * wrote (out needs room for at least that many: SCORE_E_BADARG otherwise).  Host only. */
int score_abi_struct_sizes(int64_t* out, int32_t n);

/* ---- "next" row f1: batch assembly on the device -------------------------------------- */

/* In-memory temporal bipartite graph (replaces the MongoDB documents {uid|iid,'1hop','2hop'} of
 * graph_storage.py:78-246).  CSR over (entity, time slice): the neighbours of 0-based entity e in
 * slice t are nbr[off[e*S + t] .. off[e*S + t + 1]).  Ids follow feateng_tmall.py:72-101: users
 * 1..U, items U+1..U+I.  user_rows [U, Fu] / item_rows [I, Fi] hold [id, side features...]. */
typedef struct {
  const int64_t* user_off1; const int32_t* user_nbr1;   /* user -> items it interacted with   */
  const int64_t* user_off2; const int32_t* user_nbr2;   /* user -> co-interacting users        */
  const int64_t* item_off1; const int32_t* item_nbr1;   /* item -> users                       */
  const int64_t* item_off2; const int32_t* item_nbr2;   /* item -> co-interacted items         */
  const int32_t* user_rows; const int32_t* item_rows;
  int32_t n_users, n_items, time_slice_num, user_fnum, item_fnum;
  /* GraphHandler mode (graph_loader.py:243-247, set per data set at train_score.py:295-301): 0 = 'rs', 2-hop neighbours
   * drawn uniformly (:192); 1 = 'is', drawn with probability softmax_j(1 / (degree_j - 1)) (:118-120), degree_j being
   * the slice degree of the 1-hop neighbour that 2-hop entry j was reached through (the 'degrees' lists of
   * graph_storage.py:172-176).  user_deg2 / item_deg2 are aligned with user_nbr2 / item_nbr2 (mode 1 only). */
  int32_t sample_mode;
  const int32_t* user_deg2; const int32_t* item_deg2;
} score_graph_t;

typedef struct {   /* the 8-tuple of graph_loader.py:383, device int32, B = n_lines * (1 + neg) */
  int32_t* user_1hop; int32_t* user_2hop; int32_t* item_1hop; int32_t* item_2hop;
  int32_t* target_user; int32_t* target_item; int32_t* label; int32_t* length;
} score_batch_out_t;

/* GraphHandler.gen_{user,item}_history + GraphLoader.worker (graph_loader.py:169-277, 340-383, mode
 * 'rs'): uids [n_lines], iids [n_lines*(1+neg)] (positive first).  1-hop lists are truncated /
 * cyclically padded exactly as the reference does; 2-hop lists are sampled K times with replacement
 * (uniformly, or degree-weighted in mode 'is': score_graph_t.sample_mode) from a counter-based generator
 * (`seed`), so draws differ from NumPy's but have its distribution. */
int score_batch_assemble(const score_graph_t* g, const int32_t* uids, const int32_t* iids,
                         int32_t n_lines, int32_t neg_sample_num, int32_t T, int32_t K,
                         int32_t start_time, int32_t pred_time, uint64_t seed,
                         const score_batch_out_t* out, void* stream);

/* ---- "next" row f4: ranking metrics of an evaluation pass on the device ----------------------- */

/* get_ranking_quality (train_score.py:104-142) without the host round trip of the predictions:
 * pred / ids are [n_lines, per_line] (one positive in column 0 + the sampled negatives, the layout
 * model.eval's outputs arrive in, train_score.py:153-157).  The rank of a line is the position, in
 * descending-score order with np.argsort(...)[::-1]'s tie order, of the first entry whose id equals
 * the positive's.  out6 (device) = means over the lines of NDCG@5, NDCG@10, HR@1, HR@5, HR@10, MRR;
 * ranks (optional, device int32 [n_lines]) receives the 0-based ranks.  scratch: 6 * n_lines floats. */
int score_ranking_quality(const float* pred, const int32_t* ids, int64_t n_lines, int32_t per_line,
                          float* out6, int32_t* ranks, float* scratch, int64_t scratch_floats, void* stream);

/* sklearn.metrics.roc_auc_score / log_loss of an evaluation pass (train_score.py:159-160) on the device:
 * pred float32 [n] in (0,1), label int32 [n] in {0,1}.  AUC is the Mann-Whitney statistic with average
 * ranks for tied scores (what the trapezoidal ROC area equals); log-loss clips to [eps, 1-eps] with
 * eps = 2^-52 and accumulates in double.  out2 (device doubles) = {auc, logloss}; auc is NaN when only
 * one class is present.  scratch: score_auc_scratch_bytes(n). */
int score_auc_logloss(const float* pred, const int32_t* label, int64_t n, double* out2, void* scratch,
                      int64_t scratch_bytes, void* stream);
int64_t score_auc_scratch_bytes(int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* SCORE_HIP_H */
