// tf.nn.dynamic_rnn(GRUCell(H), sequence_length) recurrence (score.py:205-208) for hidden sizes whose
// recurrent weights do not fit a CU's registers (H = 256, BASELINE.json cfg-5: 786 KB per GRU).
//
// One persistent launch per direction.  A workgroup (8 waves, two per SIMD) owns 32 batch rows of one GRU for
// all T steps with the state in LDS; samples are independent, so nothing is exchanged between workgroups and no
// grid barrier is needed.  What cannot stay on the CU -- the weights -- streams from L2 every step, laid out
// beforehand in MFMA *fragment order* (gru_frag_kernel): the B operand of four consecutive
// v_mfma_f32_32x32x2_f32 steps of one 32-column tile is ONE coalesced 1-KB global_load_dwordx4 per wave, straight
// into registers (no LDS staging, no transposes, also for the backward's W^T products).  The A operand comes
// from LDS with one ds_read_b128 per four steps (K dealt to the two lane halves in contiguous runs, row stride
// H + 4: conflict-free).  Per step and workgroup: 3,072 MFMAs (20.5 us at 2.4 GHz: the fp32 matrix peak),
// 786 KB of weights from L2 (every workgroup of a side streams the same 786 KB: L2-resident), two barriers.
// The column-sliced, grid-barrier form (each workgroup keeps a weight slice in VGPRs, all rows stream through)
// fits small batches; at B = 4096 the state (4 MB per step and side) is five times the weights, so the rows are
// what a workgroup keeps.
//
// Arithmetic is that of gru.hip's kernels (exact fp32 products, v_exp/v_rcp sigmoid and tanh).
#include "common.h"
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define SMB 32                       // batch rows per workgroup
#define SNW 8                        // waves per workgroup
// accumulator element r of a lane -> row inside the 32x32 tile; the column is lane & 31
#define SACC_ROWC(r) (((r) & 3) + 8 * ((r) >> 2))          // + 4 * (lane >> 5)

#if defined(GSP_DIVRCP)
#define GS_RCP(x) __frcp_rn(x)              // correctly rounded: a ten-instruction division sequence
#else
#define GS_RCP(x) __builtin_amdgcn_rcpf(x)  // v_rcp_f32 (1 ulp)
#endif
__device__ __forceinline__ float s_sigmoid(float x) { return GS_RCP(1.0f + __expf(-x)); }
__device__ __forceinline__ float s_tanh(float x) { return 1.0f - 2.0f * GS_RCP(__expf(2.0f * x) + 1.0f); }

// out[ct][sg][lane][e] = B(k, col), k = (lane >> 5) * (K / 2) + 4 * sg + e, col = ct * 32 + (lane & 31);
// B(k, col) = W[k * ldw + col] (trans 0: h . W) or W[col * ldw + k] (trans 1: g . W^T).
__global__ void gru_frag_kernel(const float* __restrict__ W, int ldw, int K, int N, int trans, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one float4 of the output
  const int sgs = K / 8;
  if (i >= (int64_t)(N / 32) * sgs * 64) return;
  const int lane = (int)(i & 63);
  const int sg = (int)((i >> 6) % sgs);
  const int ct = (int)((i >> 6) / sgs);
  const int col = ct * 32 + (lane & 31);
  const int k0 = (lane >> 5) * (K / 2) + 4 * sg;
  float4 v;
  if (trans) {
    v = *reinterpret_cast<const float4*>(W + (int64_t)col * ldw + k0);
  } else {
    v.x = W[(int64_t)(k0 + 0) * ldw + col]; v.y = W[(int64_t)(k0 + 1) * ldw + col];
    v.z = W[(int64_t)(k0 + 2) * ldw + col]; v.w = W[(int64_t)(k0 + 3) * ldw + col];
  }
  *reinterpret_cast<float4*>(out + i * 4) = v;
}

// acc[tt] (tt < NT) += A[32 x K] (LDS, row stride lda, this lane's row / k half) . fragment-ordered B tiles
// fr[tt] points at this lane's float4 of step group 0 of its tile; consecutive step groups are 256 floats apart.
// tools/gru_stream_probe.py builds this file with one ingredient stripped at a time (wrong results, timing only)
#if defined(GSP_NOMFMA)
#define GS_MFMA(a, b, c) ([&] { f32x16 t_ = (c); t_[0] += (a) * (b); return t_; }())
#else
#define GS_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#endif
#if defined(GSP_NOBLOAD)
#define GS_BLOAD(p) make_float4(1.f, 2.f, 3.f, 4.f)
#else
#define GS_BLOAD(p) ld4_global(p)
#endif
#if defined(GSP_NOSTORE)
#define GS_STORE(lhs, v) do { if ((v) == 123.456f) lhs = (v); } while (0)
#else
#define GS_STORE(lhs, v) lhs = (v)
#endif
#if defined(GSP_NOXLOAD)
#define GS_XLOADV(e) 0.5f
#else
#define GS_XLOADV(e) (e)
#endif
// step groups (1 KB per wave and tile each) in flight: SPF1 for a product over one column tile per wave, SPF2 over
// two.  tools/gru_stream_probe.py: deeper rings (8 / 16 groups) cost registers the backward does not have and buy
// nothing -- with the MFMAs stripped the kernels stream 786 KB per step and CU at ~41 GB/s per CU (10.6 TB/s of L2
// reads chip-wide), which is the floor the matrix work overlaps with, not a latency effect.
#ifndef SPF1
#define SPF1 4
#endif
#ifndef SPF2
#define SPF2 4
#endif
// first ring of a product's B fragments: issued EARLY (before the epilogue / barrier in front of the product), so
// the product starts on operands that have already arrived
template <int NT, int SPF>
__device__ __forceinline__ void stream_prologue(float4 (&bq)[SPF][NT], const float* (&fr)[NT]) {
#pragma unroll
  for (int p = 0; p < SPF; ++p)
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) bq[p][tt] = GS_BLOAD(fr[tt] + (int64_t)p * 256);
}
template <int NT, int SGS, int SPF>
__device__ __forceinline__ void stream_matmul(f32x16 (&acc)[NT], const float* __restrict__ arow,
                                              const float* (&fr)[NT], float4 (&bq)[SPF][NT]) {
  static_assert(SGS % SPF == 0, "whole rings");
  // The loop stays rolled (one ring revolution per trip): fully unrolled, the scheduler hoists every fragment load
  // of the phase to its top and spills ~400 registers.  A ring slot is reloaded right BEHIND the MFMAs that read it
  // (sched_barrier keeps the load there) so that it lands in the slot's own registers: written as "copy the slot,
  // reload it, then use the copy", the compiler loaded into fresh registers and moved them into the slots at the
  // end of the revolution -- behind s_waitcnt vmcnt(7..0), i.e. every revolution waited out the L2 round trip of
  // the loads it had just issued and nothing streamed under the MFMAs at all.
  float4 av = *reinterpret_cast<const float4*>(arow);
#pragma unroll 1
  for (int sg0 = 0; sg0 < SGS; sg0 += SPF) {
#pragma unroll
    for (int p = 0; p < SPF; ++p) {
      const float4 avn = *reinterpret_cast<const float4*>(arow + 4 * min(sg0 + p + 1, SGS - 1));   // next group's A operand
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) acc[tt] = GS_MFMA(av.x, bq[p][tt].x, acc[tt]);
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) acc[tt] = GS_MFMA(av.y, bq[p][tt].y, acc[tt]);
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) acc[tt] = GS_MFMA(av.z, bq[p][tt].z, acc[tt]);
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) acc[tt] = GS_MFMA(av.w, bq[p][tt].w, acc[tt]);
      __builtin_amdgcn_sched_barrier(0);
      // (the last revolution re-reads the final ring: a valid address, never used)
      const int nsg = min(sg0 + SPF + p, SGS - 1);
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) bq[p][tt] = GS_BLOAD(fr[tt] + (int64_t)nsg * 256);
      __builtin_amdgcn_sched_barrier(0);
      av = avn;
    }
  }
}

// ---------------------------------------------------------------------------------------------- addressing
// Every global access of the time loops is  base pointer + uniform byte offset (time step, accumulator row r,
// column tile: scalar registers) + ONE 32-bit per-lane byte offset fixed for the launch (lane half's rows, lane's
// column).  Per-row 64-bit pointers -- and equally the per-array, per-row induction variables loop strength
// reduction makes of any affine per-lane address -- cost 100+ VGPRs here, spilt, and every reload inside the time
// loop is an s_waitcnt vmcnt(0) that also drains the weight ring.  readfirstlane pins the uniform part.
// Batches that are not whole 32-row tiles clamp the row (rows past the batch read the last one's, never stored):
// the clamped offset of either lane half is uniform too, a lane selects its half's.
typedef __attribute__((address_space(1))) float gs_gfloat;
__device__ __forceinline__ uint64_t gs_uni(const void* base, int64_t uni) {
  const uint64_t p = reinterpret_cast<uint64_t>(base) + (uint64_t)uni;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
  return ((uint64_t)hi << 32) | lo;
}
struct GsRows {            // rows of this lane's 16 accumulator elements inside its workgroup's 32
  int kh, nv, T;
  // address of (accumulator row r, time step t, column col0 + j) in an array of row stride `stride` floats
  template <bool FULL> __device__ __forceinline__ uint64_t at(const void* base, int r, int t, int stride, int col0, int j) const {
    if (FULL) {            // row and step in the uniform part; the lane part is fixed for the launch
      return gs_uni(base, ((int64_t)(SACC_ROWC(r) * T + t) * stride + col0) * 4) + ((uint32_t)(4 * kh * T) * (uint32_t)stride + (uint32_t)j) * 4u;
    } else {               // clamped row of either lane half as two uniform offsets, selected per lane
      const uint32_t a0 = (uint32_t)((min(SACC_ROWC(r), nv - 1) * T + t) * stride) * 4u;
      const uint32_t a1 = (uint32_t)((min(4 + SACC_ROWC(r), nv - 1) * T + t) * stride) * 4u;
      return gs_uni(base, (int64_t)col0 * 4) + ((kh ? a1 : a0) + (uint32_t)j * 4u);
    }
  }
};
#define GS_LD(addr) GS_XLOADV(*(const gs_gfloat*)(addr))
#define GS_ST(addr, v) GS_STORE(*(gs_gfloat*)(addr), (v))

// ---------------------------------------------------------------------------------------------- forward
// frag: per side [WgF (2H*H) | WcF (H*H) | WcTF (H*H) | WgTF (2H*H)] floats, side stride 6*H*H.
// Wave w owns the reset, update and candidate tiles of columns 32w..32w+31, so a lane owns the same 16 (row, column)
// elements of r, u, c and h in every phase: the state and the update gate live in registers, LDS only holds the two
// MFMA A operands (h and r*h).
template <int H, bool FULL>
__global__ __launch_bounds__(64 * SNW) void gru_fwd_stream_kernel(const GruArgs a, const float* __restrict__ frag) {
  constexpr int LD = H + 4;
  static_assert(2 * H / 32 == 2 * SNW && H / 32 == SNW, "wave w owns gate tiles w, w + SNW and candidate tile w");
  constexpr int SGS = H / 8;                       // K = H: two lane halves x SGS step groups x 4 steps
  __shared__ float hs[SMB * LD], rhs[SMB * LD];
  __shared__ int lens[SMB];
  const int tiles_b = (a.B + SMB - 1) / SMB;
  const int side = blockIdx.x / tiles_b;
  const int b0 = (blockIdx.x - side * tiles_b) * SMB;
  const GruSide& sd = a.s[side];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int T = a.T;
  const float* __restrict__ WgF = frag + (int64_t)side * 6 * H * H;
  const float* __restrict__ WcF = WgF + 2 * H * H;

  for (int e = tid; e < SMB * LD; e += 64 * SNW) hs[e] = 0.f;
  if (tid < SMB) lens[tid] = (b0 + tid < a.B) ? a.length[b0 + tid] : 0;
  const GsRows rw = {kh, min(SMB, a.B - b0), T};
  unsigned rokm = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) rokm |= (FULL || 4 * kh + SACC_ROWC(r) < rw.nv) ? (1u << r) : 0u;
  const int j = wave * 32 + li;                    // this lane's column of r, u, c and h
  const int ldo = sd.ldo;
  const float* xp = sd.xproj + (int64_t)b0 * T * 3 * H;
  float* gp = sd.gates + (int64_t)b0 * T * 3 * H;
  float* op = sd.out + (int64_t)b0 * T * ldo;
  const float* fg[2] = {WgF + ((int64_t)wave * SGS * 64 + lane) * 4, WgF + ((int64_t)(wave + SNW) * SGS * 64 + lane) * 4};
  const float* fc[1] = {WcF + ((int64_t)wave * SGS * 64 + lane) * 4};
  const float* arow_h = hs + li * LD + kh * (H / 2);
  const float* arow_rh = rhs + li * LD + kh * (H / 2);
  float h[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) h[r] = 0.f;
  float4 bqg[SPF2][2], bqc[SPF1][1];
  stream_prologue<2, SPF2>(bqg, fg);
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    unsigned livem = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) livem |= (t < lens[4 * kh + SACC_ROWC(r)]) ? (1u << r) : 0u;
    // ---- gates = sigmoid(xproj[:, :2H] + h . Wg)
    float xg[2][16];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        xg[g][r] = GS_LD(rw.at<FULL>(xp, r, t, 3 * H, g * H, j));
    f32x16 acc[2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    stream_matmul<2, SGS, SPF2>(acc, arow_h, fg, bqg);
    stream_prologue<1, SPF1>(bqc, fc);            // the candidate product's first fragments: under this epilogue and the barrier
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = 4 * kh + SACC_ROWC(r);
      const float g = s_sigmoid(acc[0][r] + xg[0][r]);
      if (rokm & (1u << r)) GS_ST(rw.at<FULL>(gp, r, t, 3 * H, 0, j), g);
      rhs[i * LD + j] = g * h[r];
    }
    __syncthreads();
    // ---- c = tanh(xproj[:, 2H:] + (r*h) . Wc) ; h' = u*h + (1-u)*c.  The update gate's epilogue is not on the
    // r -> r*h -> barrier chain: it runs behind the barrier, under the candidate product's first operands.
    float xc[16], u[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      xc[r] = GS_LD(rw.at<FULL>(xp, r, t, 3 * H, 2 * H, j));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      u[r] = s_sigmoid(acc[1][r] + xg[1][r]);
      if (rokm & (1u << r)) GS_ST(rw.at<FULL>(gp, r, t, 3 * H, H, j), u[r]);
    }
    f32x16 acc2[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[0][r] = 0.f;
    stream_matmul<1, SGS, SPF1>(acc2, arow_rh, fc, bqc);
    stream_prologue<2, SPF2>(bqg, fg);            // the next step's gate product
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = 4 * kh + SACC_ROWC(r);
      const float c = s_tanh(acc2[0][r] + xc[r]);
      const float hn = u[r] * h[r] + (1.0f - u[r]) * c;
      const bool live = (livem >> r) & 1u;
      if (rokm & (1u << r)) {
        GS_ST(rw.at<FULL>(gp, r, t, 3 * H, 2 * H, j), c);
        GS_ST(rw.at<FULL>(op, r, t, ldo, 0, j), (live ? hn : 0.f));   // dynamic_rnn: zero output past the length
      }
      h[r] = live ? hn : h[r];                            // ... and the state is carried through
      hs[i * LD + j] = h[r];
    }
    __syncthreads();
  }
  if (sd.final_state) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (rokm & (1u << r)) sd.final_state[(int64_t)(b0 + 4 * kh + SACC_ROWC(r)) * H + j] = h[r];
  }
}

// ---------------------------------------------------------------------------------------------- backward
//   dh      = dout_t (live) + dh_next
//   du = dh*(h_prev - c), dc = dh*(1-u), dh_prev = dh*u
//   dpc = dc*(1-c^2);  d(rh) = dpc.Wc^T;  dr = d(rh)*h_prev;  dh_prev += d(rh)*r
//   dpr = dr*r*(1-r);  dpu = du*u*(1-u);  dh_prev += [dpr,dpu].Wg^T
// A thread owns the same (row, column) elements in all three phases (the accumulator layout of its wave's one
// column tile), so the running dL/dh lives in 16 registers and the saved activations of a step are read once.
template <int H, bool FULL>
__global__ __launch_bounds__(64 * SNW) void gru_bwd_stream_kernel(const GruArgs a, const float* __restrict__ frag) {
  constexpr int LD = H + 4, LD2 = 2 * H + 4;
  constexpr int NT = H / 32;
  static_assert(NT == SNW, "one column tile per wave");
  __shared__ float dpc[SMB * LD], dpg[SMB * LD2];
  __shared__ int lens[SMB];
  const int tiles_b = (a.B + SMB - 1) / SMB;
  const int side = blockIdx.x / tiles_b;
  const int b0 = (blockIdx.x - side * tiles_b) * SMB;
  const GruSide& sd = a.s[side];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int T = a.T;
  const float* __restrict__ WcTF = frag + (int64_t)side * 6 * H * H + 3 * H * H;
  const float* __restrict__ WgTF = WcTF + H * H;
  if (tid < SMB) lens[tid] = (b0 + tid < a.B) ? a.length[b0 + tid] : 0;
  const int j = wave * 32 + li;
  const GsRows rw = {kh, min(SMB, a.B - b0), T};
  unsigned rokm = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) rokm |= (FULL || 4 * kh + SACC_ROWC(r) < rw.nv) ? (1u << r) : 0u;
  const int ldo = sd.ldo, lddo = sd.lddo;
  const float* gp = sd.gates + (int64_t)b0 * T * 3 * H;
  const float* op = sd.out + (int64_t)b0 * T * ldo;
  const float* dop = sd.dout + (int64_t)b0 * T * lddo;
  float* dxp = sd.dxproj + (int64_t)b0 * T * 3 * H;
  float* rhp = sd.rh + (int64_t)b0 * T * H;
  float* hpp = sd.hprev + (int64_t)b0 * T * H;
  const float* fc[1] = {WcTF + ((int64_t)wave * (H / 8) * 64 + lane) * 4};
  const float* fg[1] = {WgTF + ((int64_t)wave * (2 * H / 8) * 64 + lane) * 4};
  const float* arow_c = dpc + li * LD + kh * (H / 2);
  const float* arow_g = dpg + li * LD2 + kh * H;

  float dh[16];
#pragma unroll
  for (int r = 0; r < 16; ++r)
    dh[r] = (sd.dfinal && (rokm & (1u << r))) ? sd.dfinal[(int64_t)(b0 + 4 * kh + SACC_ROWC(r)) * H + j] : 0.f;
  float n_u[16], n_c[16], n_hp[16], n_do[16];
  auto prefetch = [&](int t) {
    const int tc = max(t, 0), tp = max(t - 1, 0);          // (h_prev of t = 0 is never used)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      n_u[r] = GS_LD(rw.at<FULL>(gp, r, tc, 3 * H, H, j));
      n_c[r] = GS_LD(rw.at<FULL>(gp, r, tc, 3 * H, 2 * H, j));
      n_hp[r] = GS_LD(rw.at<FULL>(op, r, tp, ldo, 0, j));
      n_do[r] = GS_LD(rw.at<FULL>(dop, r, tc, lddo, 0, j));
    }
  };
  prefetch(T - 1);
  float4 bqc[SPF1][1], bqg[SPF1][1];
  __syncthreads();

  for (int t = T - 1; t >= 0; --t) {
    stream_prologue<1, SPF1>(bqc, fc);       // phase 2's first fragments arrive under phase 1
    float c_hp[16];
    unsigned livem = 0;
    // ---- phase 1 (elementwise): dpu, dpc ; dh <- dh_tot * u
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = 4 * kh + SACC_ROWC(r);
      const bool live = t < lens[i];
      livem |= live ? (1u << r) : 0u;
      const float u = live ? n_u[r] : 0.f, c = live ? n_c[r] : 0.f;
      c_hp[r] = (live && t > 0) ? n_hp[r] : 0.f;
      const float d = dh[r] + (live ? n_do[r] : 0.f);
      const float du = d * (c_hp[r] - c), dc = d * (1.0f - u);
      const float v_dpu = live ? du * u * (1.0f - u) : 0.f;
      const float v_dpc = live ? dc * (1.0f - c * c) : 0.f;
      dh[r] = live ? d * u : dh[r];
      if (rokm & (1u << r)) {
        GS_ST(rw.at<FULL>(hpp, r, t, H, 0, j), c_hp[r]);
        GS_ST(rw.at<FULL>(dxp, r, t, 3 * H, H, j), v_dpu);
        GS_ST(rw.at<FULL>(dxp, r, t, 3 * H, 2 * H, j), v_dpc);
      }
      dpc[i * LD + j] = v_dpc;
      dpg[i * LD2 + H + j] = v_dpu;
    }
    __syncthreads();
    // ---- phase 2: d(rh) = dpc . Wc^T ; dpr = d(rh)*h_prev*r(1-r) ; dh += d(rh)*r
    {
      float c_r[16];                                 // the reset gate of this step: arrives under the product
#pragma unroll
      for (int r = 0; r < 16; ++r) c_r[r] = GS_LD(rw.at<FULL>(gp, r, t, 3 * H, 0, j));
      f32x16 acc[1];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
      stream_matmul<1, H / 8, SPF1>(acc, arow_c, fc, bqc);
      stream_prologue<1, SPF1>(bqg, fg);           // phase 3's under this epilogue and the barrier
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = 4 * kh + SACC_ROWC(r);
        const bool live = (livem >> r) & 1u;
        const float rr = live ? c_r[r] : 0.f, hp = c_hp[r];       // both 0 past the length
        const float drh = acc[0][r];
        const float v_dpr = live ? drh * hp * rr * (1.0f - rr) : 0.f;
        dh[r] += live ? drh * rr : 0.f;
        if (rokm & (1u << r)) {
          GS_ST(rw.at<FULL>(dxp, r, t, 3 * H, 0, j), v_dpr);
          GS_ST(rw.at<FULL>(rhp, r, t, H, 0, j), (rr * hp));
        }
        dpg[i * LD2 + j] = v_dpr;
      }
    }
    __syncthreads();
    // ---- phase 3: dh += [dpr | dpu] . Wg^T ; the saved activations of step t-1 arrive under it
    prefetch(t - 1);
    {
      f32x16 acc[1];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
      stream_matmul<1, 2 * H / 8, SPF1>(acc, arow_g, fg, bqg);
#pragma unroll
      for (int r = 0; r < 16; ++r) dh[r] += acc[0][r];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- launchers
bool score_gru_stream_ok(int H) { return H == 256; }
int64_t score_gru_stream_tmp_floats(int H, int nsides) { return (int64_t)nsides * 6 * H * H; }

static int launch_frag(const float* W, int ldw, int K, int N, int trans, float* out, hipStream_t s) {
  const int64_t n4 = (int64_t)(N / 32) * (K / 8) * 64;
  hipLaunchKernelGGL(gru_frag_kernel, dim3((unsigned)cdiv64(n4, 256)), dim3(256), 0, s, W, ldw, K, N, trans, out);
  SCORE_CHECK_LAUNCH();
  return 0;
}

int score_gru_fwd_stream(GruArgs& a, int nsides, hipStream_t s) {
  const int H = a.H;
  if (!score_gru_stream_ok(H) || !a.tmp || a.tmp_floats < score_gru_stream_tmp_floats(H, nsides)) return SCORE_E_SHAPE;
  for (int i = 0; i < nsides; ++i) {
    float* base = a.tmp + (int64_t)i * 6 * H * H;
    SCORE_TRY(launch_frag(a.s[i].Wg, a.s[i].ldwg, H, 2 * H, 0, base, s));
    SCORE_TRY(launch_frag(a.s[i].Wc, a.s[i].ldwc, H, H, 0, base + 2 * H * H, s));
  }
  dim3 grid(nsides * ((a.B + SMB - 1) / SMB));
  if (a.B % SMB == 0) hipLaunchKernelGGL((gru_fwd_stream_kernel<256, true>), grid, dim3(64 * SNW), 0, s, a, a.tmp);
  else hipLaunchKernelGGL((gru_fwd_stream_kernel<256, false>), grid, dim3(64 * SNW), 0, s, a, a.tmp);
  SCORE_CHECK_LAUNCH();
  return 0;
}

int score_gru_bwd_stream(GruArgs& a, int nsides, hipStream_t s) {
  const int H = a.H;
  if (!score_gru_stream_ok(H) || !a.tmp || a.tmp_floats < score_gru_stream_tmp_floats(H, nsides)) return SCORE_E_SHAPE;
  for (int i = 0; i < nsides; ++i) {
    float* base = a.tmp + (int64_t)i * 6 * H * H;
    // d(rh) = dpc . Wc^T : B(k, col) = Wc[col][k], K = H;  dh += [dpr|dpu] . Wg^T : B(k, col) = Wg[col][k], K = 2H
    SCORE_TRY(launch_frag(a.s[i].Wc, a.s[i].ldwc, H, H, 1, base + 3 * H * H, s));
    SCORE_TRY(launch_frag(a.s[i].Wg, a.s[i].ldwg, 2 * H, H, 1, base + 4 * H * H, s));
  }
  dim3 grid(nsides * ((a.B + SMB - 1) / SMB));
  if (a.B % SMB == 0) hipLaunchKernelGGL((gru_bwd_stream_kernel<256, true>), grid, dim3(64 * SNW), 0, s, a, a.tmp);
  else hipLaunchKernelGGL((gru_bwd_stream_kernel<256, false>), grid, dim3(64 * SNW), 0, s, a, a.tmp);
  SCORE_CHECK_LAUNCH();
  return 0;
}
