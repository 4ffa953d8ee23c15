// One training step in ONE call of the host (round 5).
//
// model.train (score.py:101-116) is one sess.run per step in the reference; here it is a forward pass, a backward pass, the
// optimizer's three pieces and the next batch's index plan, on three streams tied by a dozen events.  Queued call by call from
// Python, the bookkeeping between the calls (which event, which stream, what is pending) costs as much host time as the calls
// themselves at the reference's own batch sizes (tools/host_calls.py: ~85 us of "python + torch" against ~110 us inside the
// library for a step whose dependent device work is ~150 us).  score_train_step makes the steady-state sequence here, from a
// struct the caller fills once and touches up per step.  It is the SAME sequence of the same entry points with the same
// arguments on the same streams as score_amd/model.py's call-by-call path (which remains: first steps, evaluation in between,
// stage events, wrong hints -- anything not steady state), so the two are interchangeable step by step, bit for bit
// (tests/test_gpu_persample.py).
#include <cstddef>
#include "common.h"

#define HIPTRY_(expr)                                  \
  do {                                                 \
    hipError_t e__ = (expr);                           \
    if (e__ != hipSuccess) return (int)e__;            \
  } while (0)

extern "C" int score_train_step(const score_config_t* cfg, const score_state_t* st_in, const score_batch_t* batch,
                                const score_train_step_t* p, void* stream) {
  if (!cfg || !st_in || !batch || !p || !p->table || !p->w || !p->w_m || !p->w_v || !p->w_g || !p->side_stream ||
      !p->ev_b4 || !p->ev_grads || !p->ev_sweep || !p->ev_plan || !p->ev_ahead)
    return SCORE_E_BADARG;
  hipStream_t s = (hipStream_t)stream, side = (hipStream_t)p->side_stream;
  score_state_t st = *st_in;
  // (0) side stream, FIRST: the NEXT batch's index plan into ITS workspace (next_workspace: not the one this step computes in --
  //     the caller alternates two).  It depends on the ids only, and nothing of this step touches that workspace, so the sort runs
  //     beside this step's passes instead of behind its scatter: pull(n) -> look-ahead -> sort(n + 1) -> pull(n + 1) was the
  //     longest cycle of the step at the Taobao / Tmall default shapes (the sort is ~100 us).  The last readers of that workspace's
  //     plan (the scatter two steps back) are behind an event this stream waited for a step ago.
  if (p->next_batch && p->ev_plan_next) {
    // (on a stream of its own when the caller has one: the look-ahead catch-up below must not queue behind ~100 us of sort;
    //  ev_b4 still holds the PREVIOUS call's record here -- that step's scatter, the last launch that touched this side of
    //  the caller's two workspaces but one)
    hipStream_t ps = p->plan_stream ? (hipStream_t)p->plan_stream : side;
    if (p->plan_stream) HIPTRY_(hipStreamWaitEvent(ps, (hipEvent_t)p->ev_b4, 0));
    score_state_t sp = *st_in;
    sp.workspace = p->next_workspace; sp.workspace_bytes = p->next_workspace_bytes;
    sp.id_status = nullptr;        // (the ids are reported by the forward pass of that batch: the word guards THIS step's optimizer)
    sp.gather_done_event = sp.plan_done_event = sp.grads_done_event = sp.loss_done_event = nullptr;
    sp.loss_host = nullptr; sp.plan_workspace = nullptr;
    SCORE_TRY(score_index_plan(cfg, &sp, p->next_batch, 1, 0, (void*)ps));
    HIPTRY_(hipEventRecord((hipEvent_t)p->ev_plan_next, ps));
  }
  // (1) what the previous step left running on the side stream on rows this batch reads: the look-ahead catch-up of exactly these
  //     rows, the window slice
  if (p->wait_ahead) HIPTRY_(hipStreamWaitEvent(s, (hipEvent_t)p->ev_ahead, 0));
  if (p->wait_sweep) HIPTRY_(hipStreamWaitEvent(s, (hipEvent_t)p->ev_sweep, 0));
  // (2) forward (ev_loss: only somebody who reads the loss from the host needs an event behind the forward kernel)
  st.gather_done_event = nullptr;
  st.loss_done_event = p->loss_host ? p->ev_loss : nullptr;
  st.loss_host = p->loss_host;
  st.plan_done_event = nullptr;
  st.grads_done_event = nullptr;
  SCORE_TRY(score_forward(cfg, &st, batch, p->reg_lambda, p->keep_prob, nullptr, nullptr, p->drop_seed, p->fwd_stage_events, stream));
  // (3) backward: the row scatter behind this batch's index plan (sorted by the previous call), the dense gradient's finishers on
  //     the context's side stream
  st.plan_done_event = p->ev_plan;
  st.grads_done_event = p->ev_grads;
  void* ev[6] = {nullptr, nullptr, nullptr, nullptr, p->ev_b4, nullptr};
  SCORE_TRY(score_backward(cfg, &st, batch, p->keep_prob, p->w_g, const_cast<float*>(p->table->g), ev, stream));
  // (4) the step's ApplyAdam: rows with a gradient and the dense variables in one launch, behind the finishers -- queued BEFORE the
  //     side stream's work: the next forward pass waits for this launch (through the weight images) longer than for the look-ahead
  {
    HIPTRY_(hipStreamWaitEvent(s, (hipEvent_t)p->ev_grads, 0));
    SCORE_TRY(score_adam_touched_and_dense(p->table, p->step, p->alpha, p->w, p->w_m, p->w_v, p->w_g, p->n_w, p->n_reg, p->reg_lambda,
                                           p->skipped, stream));
  }
  // (5) side stream, behind the row scatter: the NEXT batch's rows brought up to date through this step, then its index plan
  HIPTRY_(hipStreamWaitEvent(side, (hipEvent_t)p->ev_b4, 0));
  if (p->next_batch) {
    SCORE_TRY(score_adam_catchup_ids_through(p->table, p->next_ids, p->n_next_ids, p->step, p->alpha, p->side_stream));
    HIPTRY_(hipEventRecord((hipEvent_t)p->ev_ahead, side));
    if (!p->ev_plan_next) {        // (one workspace, one plan event: the plan behind this step's scatter, as before)
      score_state_t sp = *st_in;
      sp.workspace = p->next_workspace; sp.workspace_bytes = p->next_workspace_bytes;
      sp.id_status = nullptr;
      sp.gather_done_event = sp.plan_done_event = sp.grads_done_event = sp.loss_done_event = nullptr;
      sp.loss_host = nullptr; sp.plan_workspace = nullptr;
      SCORE_TRY(score_index_plan(cfg, &sp, p->next_batch, 1, 0, p->side_stream));
      HIPTRY_(hipEventRecord((hipEvent_t)p->ev_plan, side));
    }
  }
  // (6) ... and LAST on the side stream this step's slice of the table (rows lagging behind step - 1, none of them this batch's or
  //     -- any more -- the next one's): nothing of the next step waits for it (its own side-stream work queues behind it, its
  //     forward pass reads rows the slice skips, its touched-row update publishes counts the slice cannot mistake: adam_tiled.hip);
  //     ev_sweep is for whoever reads the table otherwise (a flush, an evaluation, the call-by-call path)
  if (p->slice_hi > p->slice_lo) {
    SCORE_TRY(score_adam_catchup_rows(p->table, p->slice_lo, p->slice_hi, p->slice_upto, p->side_stream));
    HIPTRY_(hipEventRecord((hipEvent_t)p->ev_sweep, side));
  }
  return 0;
}

// sizeof / a late field's offset of every struct of include/score_hip.h, in the header's order: what a binding written in another
// language (score_amd/_lib.py's ctypes structures) checks itself against before the first call (tests/test_abi.py)
extern "C" int score_abi_struct_sizes(int64_t* out, int32_t n) {
  const int64_t v[] = {(int64_t)sizeof(score_step_scalars_t), (int64_t)sizeof(score_config_t), (int64_t)sizeof(score_param_entry_t),
                       (int64_t)sizeof(score_batch_t), (int64_t)sizeof(score_workspace_t), (int64_t)sizeof(score_guard_t),
                       (int64_t)sizeof(score_adam_table_t), (int64_t)sizeof(score_state_t), (int64_t)sizeof(score_train_step_t),
                       (int64_t)sizeof(score_graph_t), (int64_t)sizeof(score_batch_out_t),
                       (int64_t)offsetof(score_state_t, plan_workspace), (int64_t)offsetof(score_train_step_t, plan_stream),
                       (int64_t)offsetof(score_adam_table_t, skipped_steps)};
  const int32_t have = (int32_t)(sizeof(v) / sizeof(v[0]));
  if (!out || n < have) return SCORE_E_BADARG;
  for (int32_t i = 0; i < have; ++i) out[i] = v[i];
  return have;
}
