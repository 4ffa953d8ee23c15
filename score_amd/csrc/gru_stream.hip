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

__device__ __forceinline__ float s_sigmoid(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float s_tanh(float x) { return 1.0f - 2.0f * __frcp_rn(__expf(2.0f * x) + 1.0f); }

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
#define GS_XLOAD(p) 0.5f
#else
#define GS_XLOAD(p) ld1_global(p)
#endif
#define SPF 4                       // step groups in flight: ~1.5 k cycles of MFMA work cover an L2 round trip
// first ring of a product's B fragments: issued EARLY (before the epilogue / barrier in front of the product), so
// the product starts on operands that have already arrived
template <int NT>
__device__ __forceinline__ void stream_prologue(float4 (&bq)[SPF][NT], const float* (&fr)[NT]) {
#pragma unroll
  for (int p = 0; p < SPF; ++p)
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) bq[p][tt] = GS_BLOAD(fr[tt] + (int64_t)p * 256);
}
template <int NT, int SGS>
__device__ __forceinline__ void stream_matmul(f32x16 (&acc)[NT], const float* __restrict__ arow,
                                              const float* (&fr)[NT], float4 (&bq)[SPF][NT]) {
  static_assert(SGS % SPF == 0, "whole rings");
  // the loop stays rolled (one ring revolution per trip): fully unrolled, the scheduler hoists every fragment load
  // of the phase to its top and spills ~400 registers
#pragma unroll 1
  for (int sg0 = 0; sg0 < SGS; sg0 += SPF) {
#pragma unroll
    for (int p = 0; p < SPF; ++p) {
      const float4 av = *reinterpret_cast<const float4*>(arow + 4 * (sg0 + p));
      float4 bc[NT];
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        bc[tt] = bq[p][tt];
        // (the last revolution re-reads the final ring: a valid address, never used)
        const int nsg = min(sg0 + SPF + p, SGS - 1);
        bq[p][tt] = GS_BLOAD(fr[tt] + (int64_t)nsg * 256);
      }
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) acc[tt] = GS_MFMA(av.x, bc[tt].x, acc[tt]);
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) acc[tt] = GS_MFMA(av.y, bc[tt].y, acc[tt]);
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) acc[tt] = GS_MFMA(av.z, bc[tt].z, acc[tt]);
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) acc[tt] = GS_MFMA(av.w, bc[tt].w, acc[tt]);
    }
  }
}

// ---------------------------------------------------------------------------------------------- forward
// frag: per side [WgF (2H*H) | WcF (H*H) | WcTF (H*H) | WgTF (2H*H)] floats, side stride 6*H*H.
template <int H, bool FULL>
__global__ __launch_bounds__(64 * SNW) void gru_fwd_stream_kernel(const GruArgs a, const float* __restrict__ frag) {
  constexpr int LD = H + 4;
  constexpr int NTG = 2 * H / 32, NTC = H / 32;
  constexpr int TGW = NTG / SNW, TCW = NTC / SNW;
  static_assert(NTG % SNW == 0 && NTC % SNW == 0, "every wave owns whole column tiles");
  constexpr int SGS = H / 8;                       // K = H: two lane halves x SGS step groups x 4 steps
  __shared__ float hs[SMB * LD], rhs[SMB * LD], us[SMB * LD];
  __shared__ int lens[SMB];
  const int tiles_b = (a.B + SMB - 1) / SMB;
  const int side = blockIdx.x / tiles_b;
  const GruSide& sd = a.s[side];
  const int b0 = (blockIdx.x - side * tiles_b) * SMB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int T = a.T;
  const float* __restrict__ WgF = frag + (int64_t)side * 6 * H * H;
  const float* __restrict__ WcF = WgF + 2 * H * H;

  for (int e = tid; e < SMB * LD; e += 64 * SNW) hs[e] = 0.f;
  if (tid < SMB) lens[tid] = (b0 + tid < a.B) ? a.length[b0 + tid] : 0;
  // rows of this lane's accumulator elements: row 4*kh + SACC_ROWC(r) of the workgroup's 32; past the batch ->
  // clamped (never stored).  Addresses = one uniform base pointer per array + 32-bit element offsets (64-bit
  // pointers per row cost 96 registers and spilled)
  const int nv = min(SMB, a.B - b0);
  unsigned rokm = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) rokm |= (FULL || 4 * kh + SACC_ROWC(r) < nv) ? (1u << r) : 0u;
  auto rt = [&](int r) -> int {            // (row inside the workgroup) * T
    const int i = FULL ? 4 * kh + SACC_ROWC(r) : min(4 * kh + SACC_ROWC(r), nv - 1);
    return i * T;
  };
  const float* __restrict__ xp = sd.xproj + (int64_t)b0 * T * 3 * H;
  float* __restrict__ gp = sd.gates + (int64_t)b0 * T * 3 * H;
  float* __restrict__ op = sd.out + (int64_t)b0 * T * sd.ldo;
  const int ldo = sd.ldo;
  const float* fg[TGW];
  const float* fc[TCW];
#pragma unroll
  for (int tt = 0; tt < TGW; ++tt) fg[tt] = WgF + ((int64_t)(wave + SNW * tt) * SGS * 64 + lane) * 4;
#pragma unroll
  for (int tt = 0; tt < TCW; ++tt) fc[tt] = WcF + ((int64_t)(wave + SNW * tt) * SGS * 64 + lane) * 4;
  const float* arow_h = hs + li * LD + kh * (H / 2);
  const float* arow_rh = rhs + li * LD + kh * (H / 2);
  float4 bqg[SPF][TGW], bqc[SPF][TCW];
  stream_prologue<TGW>(bqg, fg);
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    // ---- gates = sigmoid(xproj[:, :2H] + h . Wg)
    float xg[TGW][16];
#pragma unroll
    for (int tt = 0; tt < TGW; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        xg[tt][r] = GS_XLOAD(xp + ((rt(r) + t) * 3 * H + (wave + SNW * tt) * 32 + li));
    f32x16 acc[TGW];
#pragma unroll
    for (int tt = 0; tt < TGW; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;
    stream_matmul<TGW, SGS>(acc, arow_h, fg, bqg);
    stream_prologue<TCW>(bqc, fc);          // the candidate product's first fragments: under this epilogue and the barrier
#pragma unroll
    for (int tt = 0; tt < TGW; ++tt) {
      const int tile = wave + SNW * tt;
      const int j = tile * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = 4 * kh + SACC_ROWC(r);
        const float g = s_sigmoid(acc[tt][r] + xg[tt][r]);
        if (rokm & (1u << r)) GS_STORE(gp[(rt(r) + t) * 3 * H + j], g);
        if (tile < NTG / 2) rhs[i * LD + j] = g * hs[i * LD + j];      // reset gate: columns [0, H)
        else us[i * LD + (j - H)] = g;                                 // update gate
      }
    }
    __syncthreads();
    // ---- c = tanh(xproj[:, 2H:] + (r*h) . Wc) ; h' = u*h + (1-u)*c
    float xc[TCW][16];
#pragma unroll
    for (int tt = 0; tt < TCW; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        xc[tt][r] = GS_XLOAD(xp + ((rt(r) + t) * 3 * H + 2 * H + (wave + SNW * tt) * 32 + li));
    f32x16 acc2[TCW];
#pragma unroll
    for (int tt = 0; tt < TCW; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[tt][r] = 0.f;
    stream_matmul<TCW, SGS>(acc2, arow_rh, fc, bqc);
    stream_prologue<TGW>(bqg, fg);          // the next step's gate product
#pragma unroll
    for (int tt = 0; tt < TCW; ++tt) {
      const int j = (wave + SNW * tt) * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = 4 * kh + SACC_ROWC(r);
        const float c = s_tanh(acc2[tt][r] + xc[tt][r]);
        const float u = us[i * LD + j], h = hs[i * LD + j];
        const float hn = u * h + (1.0f - u) * c;
        const bool live = t < lens[i];
        if (rokm & (1u << r)) {
          const int row = rt(r) + t;
          GS_STORE(gp[row * 3 * H + 2 * H + j], c);
          GS_STORE(op[row * ldo + j], (live ? hn : 0.f));   // dynamic_rnn: zero output past the length
        }
        hs[i * LD + j] = live ? hn : h;                   // ... and the state is carried through
      }
    }
    __syncthreads();
  }
  if (sd.final_state)
    for (int e = tid; e < SMB * H; e += 64 * SNW) {
      const int i = e / H, j = e - i * H;
      if (b0 + i < a.B) sd.final_state[(int64_t)(b0 + i) * H + j] = hs[i * LD + j];
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
  const GruSide& sd = a.s[side];
  const int b0 = (blockIdx.x - side * tiles_b) * SMB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int T = a.T;
  const float* __restrict__ WcTF = frag + (int64_t)side * 6 * H * H + 3 * H * H;
  const float* __restrict__ WgTF = WcTF + H * H;
  if (tid < SMB) lens[tid] = (b0 + tid < a.B) ? a.length[b0 + tid] : 0;
  const int j = wave * 32 + li;
  const int nv = min(SMB, a.B - b0);
  unsigned rokm = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) rokm |= (FULL || 4 * kh + SACC_ROWC(r) < nv) ? (1u << r) : 0u;
  auto rt = [&](int r) -> int {            // (row inside the workgroup) * T; 32-bit offsets from uniform base pointers
    const int i = FULL ? 4 * kh + SACC_ROWC(r) : min(4 * kh + SACC_ROWC(r), nv - 1);
    return i * T;
  };
  const float* __restrict__ gp = sd.gates + (int64_t)b0 * T * 3 * H;
  const float* __restrict__ op = sd.out + (int64_t)b0 * T * sd.ldo;
  const float* __restrict__ dop = sd.dout + (int64_t)b0 * T * sd.lddo;
  float* __restrict__ dxp = sd.dxproj + (int64_t)b0 * T * 3 * H;
  float* __restrict__ rhp = sd.rh + (int64_t)b0 * T * H;
  float* __restrict__ hpp = sd.hprev + (int64_t)b0 * T * H;
  const int ldo = sd.ldo, lddo = sd.lddo;
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
    const int tc = max(t, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rt(r) + tc;
      n_u[r] = GS_XLOAD(gp + (row * 3 * H + H + j));
      n_c[r] = GS_XLOAD(gp + (row * 3 * H + 2 * H + j));
      n_hp[r] = GS_XLOAD(op + ((row - (tc > 0 ? 1 : 0)) * ldo + j));
      n_do[r] = GS_XLOAD(dop + (row * lddo + j));
    }
  };
  prefetch(T - 1);
  float4 bqc[SPF][1], bqg[SPF][1];
  __syncthreads();

  for (int t = T - 1; t >= 0; --t) {
    stream_prologue<1>(bqc, fc);             // phase 2's first fragments arrive under phase 1
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
        const int row = rt(r) + t;
        GS_STORE(hpp[row * H + j], c_hp[r]);
        GS_STORE(dxp[row * 3 * H + H + j], v_dpu);
        GS_STORE(dxp[row * 3 * H + 2 * H + j], v_dpc);
      }
      dpc[i * LD + j] = v_dpc;
      dpg[i * LD2 + H + j] = v_dpu;
    }
    __syncthreads();
    // ---- phase 2: d(rh) = dpc . Wc^T ; dpr = d(rh)*h_prev*r(1-r) ; dh += d(rh)*r
    {
      float c_r[16];                                 // the reset gate of this step: arrives under the product
#pragma unroll
      for (int r = 0; r < 16; ++r) c_r[r] = GS_XLOAD(gp + ((rt(r) + t) * 3 * H + j));
      f32x16 acc[1];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
      stream_matmul<1, H / 8>(acc, arow_c, fc, bqc);
      stream_prologue<1>(bqg, fg);           // phase 3's under this epilogue and the barrier
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = 4 * kh + SACC_ROWC(r);
        const bool live = (livem >> r) & 1u;
        const float rr = live ? c_r[r] : 0.f, hp = c_hp[r];       // both 0 past the length
        const float drh = acc[0][r];
        const float v_dpr = live ? drh * hp * rr * (1.0f - rr) : 0.f;
        dh[r] += live ? drh * rr : 0.f;
        if (rokm & (1u << r)) {
          const int row = rt(r) + t;
          GS_STORE(dxp[row * 3 * H + j], v_dpr);
          GS_STORE(rhp[row * H + j], (rr * hp));
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
      stream_matmul<1, 2 * H / 8>(acc, arow_g, fg, bqg);
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
