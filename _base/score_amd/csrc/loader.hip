// Device-side batch assembly: replaces the per-sample Python loops of the reference loader
// (code/score/graph_loader.py: GraphHandler.gen_{user,item}_neighbor_rs :169-238, gen_*_history
// :240-277, GraphLoader.worker :340-383) -- one MongoDB find + nested-list building per entity --
// with one launch over an in-memory temporal CSR graph, emitting the int32 [B,T,K,F] tensors of
// score.py:21-30 directly in HBM.
//   1-hop list of (entity, slice): truncated to the first K, or cyclically padded to K (:178-182);
//   2-hop list: K uniform draws with replacement (np.random.choice, :192; counter-based RNG here);
//   empty list -> all-zero dummy objects (:90-91, :189, :200);
//   slices t >= pred_time-start_time replicate the last real slice, draws included (:254-256);
//   every neighbour id is expanded to its feature row [id, side features...] (:186-188);
//   a user's tensors are shared by its 1+neg candidate rows (:363-364).
#include "common.h"
#include "kernels.h"

struct AssembleSide {
  const int64_t* off1; const int32_t* nbr1; const int64_t* off2; const int32_t* nbr2;
  const int32_t* deg2;       // mode 'is': degree behind every 2-hop entry (aligned with nbr2), else null
  const int32_t* ent;        // entity ids of this side, one per slot
  const int32_t* rows1; int F1; int base1;   // feature rows of the 1-hop neighbour type; id - base1 = row
  const int32_t* rows2; int F2; int base2;
  int32_t* out1; int32_t* out2;
  int n_slots, copies, ent_base;              // ent - ent_base = 0-based entity index
};

__global__ void assemble_kernel(AssembleSide sd, int S, int T, int K, int start_time, int length, uint64_t seed) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)sd.n_slots * T * K) return;
  const int k = (int)(i % K);
  const int t = (int)((i / K) % T);
  const int slot = (int)(i / ((int64_t)K * T));
  const int e = sd.ent[slot] - sd.ent_base;
  const int ts = start_time + (t < length ? t : length - 1);      // tail slices copy the last real one
  const int64_t cs = (int64_t)e * S + ts;
  // 1-hop: first K, or cyclic pad
  int32_t id1 = 0;
  {
    const int64_t b = sd.off1[cs];
    const int len = (int)(sd.off1[cs + 1] - b);
    if (len > 0) id1 = sd.nbr1[b + (len > K ? k : k % len)];
  }
  // 2-hop: uniform draw with replacement; the draw belongs to the SOURCE slice, so replicas agree
  int32_t id2 = 0;
  {
    const int64_t b = sd.off2[cs];
    const int len = (int)(sd.off2[cs + 1] - b);
    if (len > 0) {
      const uint64_t ctr = ((uint64_t)(uint32_t)sd.ent[slot] << 20) ^ ((uint64_t)ts << 8) ^ (uint64_t)k;
      const float u = hash_uniform(seed, ctr);
      int pick;
      if (sd.deg2) {
        // mode 'is' (graph_loader.py:118-120): p_j = softmax_j(1 / (degree_j - 1)); np.random.choice(p=...) walks the
        // cumulative sum -- same here (lists hold at most max_2hop = 100 entries; degrees are >= 2 by construction)
        float tot = 0.f;
        for (int j = 0; j < len; ++j) tot += __expf(1.0f / (float)max(sd.deg2[b + j] - 1, 1));
        const float target = u * tot;
        float run = 0.f;
        pick = len - 1;
        for (int j = 0; j < len; ++j) {
          run += __expf(1.0f / (float)max(sd.deg2[b + j] - 1, 1));
          if (target < run) { pick = j; break; }
        }
      } else {
        pick = (int)(u * (float)len);
      }
      id2 = sd.nbr2[b + (pick < len ? pick : len - 1)];
    }
  }
  for (int c = 0; c < sd.copies; ++c) {
    const int64_t row = ((int64_t)(slot * sd.copies + c) * T + t) * K + k;
    for (int f = 0; f < sd.F1; ++f)
      sd.out1[row * sd.F1 + f] = id1 ? sd.rows1[(int64_t)(id1 - sd.base1) * sd.F1 + f] : 0;
    for (int f = 0; f < sd.F2; ++f)
      sd.out2[row * sd.F2 + f] = id2 ? sd.rows2[(int64_t)(id2 - sd.base2) * sd.F2 + f] : 0;
  }
}

__global__ void assemble_targets_kernel(const int32_t* __restrict__ uids, const int32_t* __restrict__ iids, int B,
                                        int copies, const int32_t* __restrict__ user_rows, int Fu,
                                        const int32_t* __restrict__ item_rows, int Fi, int n_users, int length,
                                        int32_t* __restrict__ tu, int32_t* __restrict__ ti,
                                        int32_t* __restrict__ label, int32_t* __restrict__ len_out) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int u = uids[b / copies], it = iids[b];
  for (int f = 0; f < Fu; ++f) tu[b * Fu + f] = user_rows[(int64_t)(u - 1) * Fu + f];
  for (int f = 0; f < Fi; ++f) ti[b * Fi + f] = item_rows[(int64_t)(it - n_users - 1) * Fi + f];
  label[b] = (b % copies) == 0 ? 1 : 0;      // first candidate of a line is the positive (:377-380)
  len_out[b] = length;                        // pred_time - start_time (:382)
}

extern "C" int score_batch_assemble(const score_graph_t* g, const int32_t* uids, const int32_t* iids,
                                    int32_t n_lines, int32_t neg_sample_num, int32_t T, int32_t K,
                                    int32_t start_time, int32_t pred_time, uint64_t seed,
                                    const score_batch_out_t* out, void* stream) {
  if (!g || !uids || !iids || !out || n_lines <= 0 || neg_sample_num < 0 || T <= 0 || K <= 0) return SCORE_E_BADARG;
  if (!g->user_off1 || !g->user_nbr1 || !g->user_off2 || !g->user_nbr2 || !g->item_off1 || !g->item_nbr1 ||
      !g->item_off2 || !g->item_nbr2 || !g->user_rows || !g->item_rows)
    return SCORE_E_BADARG;
  const int length = pred_time - start_time;
  if (length <= 0 || length > T || pred_time > g->time_slice_num || start_time < 0) return SCORE_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int copies = 1 + neg_sample_num, B = n_lines * copies;
  AssembleSide us;
  if (g->sample_mode != 0 && g->sample_mode != 1) return SCORE_E_BADARG;
  if (g->sample_mode == 1 && (!g->user_deg2 || !g->item_deg2)) return SCORE_E_BADARG;
  us.off1 = g->user_off1; us.nbr1 = g->user_nbr1; us.off2 = g->user_off2; us.nbr2 = g->user_nbr2;
  us.deg2 = g->sample_mode == 1 ? g->user_deg2 : nullptr;
  us.ent = uids; us.rows1 = g->item_rows; us.F1 = g->item_fnum; us.base1 = g->n_users + 1;
  us.rows2 = g->user_rows; us.F2 = g->user_fnum; us.base2 = 1;
  us.out1 = out->user_1hop; us.out2 = out->user_2hop; us.n_slots = n_lines; us.copies = copies; us.ent_base = 1;
  AssembleSide is;
  is.off1 = g->item_off1; is.nbr1 = g->item_nbr1; is.off2 = g->item_off2; is.nbr2 = g->item_nbr2;
  is.deg2 = g->sample_mode == 1 ? g->item_deg2 : nullptr;
  is.ent = iids; is.rows1 = g->user_rows; is.F1 = g->user_fnum; is.base1 = 1;
  is.rows2 = g->item_rows; is.F2 = g->item_fnum; is.base2 = g->n_users + 1;
  is.out1 = out->item_1hop; is.out2 = out->item_2hop; is.n_slots = B; is.copies = 1; is.ent_base = g->n_users + 1;
  int64_t nu = (int64_t)n_lines * T * K, ni = (int64_t)B * T * K;
  hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)cdiv64(nu, 256)), dim3(256), 0, s, us, g->time_slice_num, T, K,
                     start_time, length, seed);
  SCORE_CHECK_LAUNCH();
  hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)cdiv64(ni, 256)), dim3(256), 0, s, is, g->time_slice_num, T, K,
                     start_time, length, seed ^ 0xA5A5A5A5DEADBEEFull);
  SCORE_CHECK_LAUNCH();
  hipLaunchKernelGGL(assemble_targets_kernel, dim3((B + 255) / 256), dim3(256), 0, s, uids, iids, B, copies,
                     g->user_rows, g->user_fnum, g->item_rows, g->item_fnum, g->n_users, length, out->target_user,
                     out->target_item, out->label, out->length);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ ranking metrics (train_score.py:104-142)
// one wave per line: the positive's id may also appear among the negatives, so every entry carrying it is ranked
// and the best position wins.  Position of entry c in np.argsort(p)[::-1]: entries with a larger score, plus
// equal scores with a larger index (the stable ascending sort, reversed).
__global__ __launch_bounds__(256) void rank_lines_kernel(const float* __restrict__ pred, const int32_t* __restrict__ ids,
                                                         int64_t n_lines, int per_line, int32_t* __restrict__ ranks,
                                                         float* __restrict__ metrics) {
  const int lane = threadIdx.x & 63;
  const int64_t line = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (line >= n_lines) return;
  const float* p = pred + line * per_line;
  const int32_t* id = ids + line * per_line;
  const int32_t pos_id = id[0];
  int best = per_line;
  for (int c = lane; c < per_line; c += 64) {
    if (id[c] != pos_id) continue;
    const float pc = p[c];
    int before = 0;
    for (int j = 0; j < per_line; ++j) before += (p[j] > pc) || (p[j] == pc && j > c);
    best = min(best, before);
  }
  for (int off = 32; off > 0; off >>= 1) best = min(best, __shfl_xor(best, off, 64));
  if (lane == 0) {
    if (ranks) ranks[line] = best;
    const float gain = logf(2.0f) / logf((float)best + 2.0f);
    float* m = metrics + line * 6;
    m[0] = best < 5 ? gain : 0.f; m[1] = best < 10 ? gain : 0.f;
    m[2] = best < 1 ? 1.f : 0.f; m[3] = best < 5 ? 1.f : 0.f; m[4] = best < 10 ? 1.f : 0.f;
    m[5] = 1.0f / (float)(best + 1);
  }
}
// means of the six per-line metrics: one block, fixed order, double accumulation
__global__ __launch_bounds__(256) void rank_mean_kernel(const float* __restrict__ metrics, int64_t n_lines,
                                                        float* __restrict__ out6) {
  __shared__ double sh[256];
  for (int q = 0; q < 6; ++q) {
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n_lines; i += 256) acc += (double)metrics[i * 6 + q];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) out6[q] = (float)(sh[0] / (double)n_lines);
    __syncthreads();
  }
}

extern "C" int score_ranking_quality(const float* pred, const int32_t* ids, int64_t n_lines, int32_t per_line,
                                     float* out6, int32_t* ranks, float* scratch, int64_t scratch_floats,
                                     void* stream) {
  if (!pred || !ids || !out6 || !scratch || n_lines <= 0 || per_line <= 0) return SCORE_E_BADARG;
  if (scratch_floats < 6 * n_lines) return SCORE_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(rank_lines_kernel, dim3((unsigned)cdiv64(n_lines, 4)), dim3(256), 0, s, pred, ids, n_lines,
                     per_line, ranks, scratch);
  SCORE_CHECK_LAUNCH();
  hipLaunchKernelGGL(rank_mean_kernel, dim3(1), dim3(256), 0, s, scratch, n_lines, out6);
  SCORE_CHECK_LAUNCH();
  return 0;
}
