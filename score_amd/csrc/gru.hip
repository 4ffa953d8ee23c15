// tf.nn.dynamic_rnn(GRUCell(H), sequence_length) recurrence (score.py:205-208),
// forward and backward, as ONE persistent launch per direction: a workgroup owns
// 32 batch rows (samples are independent) and walks all T steps with the hidden
// state in LDS; the two small products of a step (h.Wg[H,2H], (r*h).Wc[H,H]) run
// on v_mfma_f32_32x32x2_f32 (exact fp32), weights stream from L2.
//
// GRUCell (TF 1.x): [r,u] = sigmoid([x,h].Wg + bg) (r first), c = tanh([x,r*h].Wc + bc),
// h' = u*h + (1-u)*c.  The x-part (x.Wx + b) is hoisted into one big GEMM: `xproj`.
#include "common.h"
#include "kernels.h"

#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define RB 32
static bool gru_reg_ok(int H);
int score_gru_fwd_multi(GruArgs& a, int nsides, hipStream_t s);
int score_gru_bwd_multi(GruArgs& a, int nsides, hipStream_t s);  // batch rows per workgroup

// acc(32 x 32 cols starting at j0) = Ash[32][K] (LDS, row stride lds_ld) . Wm[K][ldw] (global)
__device__ __forceinline__ f32x16 tile_matmul(const float* __restrict__ Ash, int lds_ld, int Kdim,
                                              const float* __restrict__ Wm, int ldw, int j0, int ncols,
                                              int lane) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int i = lane & 31, kh = lane >> 5;
  const int j = j0 + i;
  const bool jok = j < ncols;
  const float* wp = Wm + (jok ? j : 0);
  int k = 0;
  for (; k + 8 <= Kdim; k += 8) {  // 4 MFMAs per trip, loads issued together
    float a0 = Ash[i * lds_ld + k + kh], a1 = Ash[i * lds_ld + k + 2 + kh];
    float a2 = Ash[i * lds_ld + k + 4 + kh], a3 = Ash[i * lds_ld + k + 6 + kh];
    float b0 = jok ? wp[(int64_t)(k + kh) * ldw] : 0.f, b1 = jok ? wp[(int64_t)(k + 2 + kh) * ldw] : 0.f;
    float b2 = jok ? wp[(int64_t)(k + 4 + kh) * ldw] : 0.f, b3 = jok ? wp[(int64_t)(k + 6 + kh) * ldw] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3, acc, 0, 0, 0);
  }
  for (; k < Kdim; k += 2) {
    int kk = k + kh;
    float a = kk < Kdim ? Ash[i * lds_ld + kk] : 0.f;
    float b = (jok && kk < Kdim) ? wp[(int64_t)kk * ldw] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  return acc;
}
// accumulator element r of lane -> (row, col) inside the 32x32 tile
#define ACC_ROW(r, lane) (((r) & 3) + 8 * ((r) >> 2) + 4 * ((lane) >> 5))
#define ACC_COL(lane) ((lane) & 31)

__global__ __launch_bounds__(256) void gru_fwd_kernel(int B, int T, int H, const float* __restrict__ xproj,
                                                      const float* __restrict__ Wg, int ldwg,
                                                      const float* __restrict__ Wc, int ldwc,
                                                      const int32_t* __restrict__ length,
                                                      float* __restrict__ out, int ldo,
                                                      float* __restrict__ gates, float* __restrict__ final_state) {
  extern __shared__ float sm[];
  const int ld = H + 1;
  float* hs = sm;             // [RB][H+1] hidden state
  float* rhs = hs + RB * ld;  // r * h
  float* us = rhs + RB * ld;  // update gate
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * RB;
  for (int e = tid; e < RB * ld; e += blockDim.x) hs[e] = 0.f;
  __syncthreads();
  const int ntg = (2 * H + 31) / 32, ntc = (H + 31) / 32;
  for (int t = 0; t < T; ++t) {
    // gates = sigmoid(xproj[:, 0:2H] + h.Wg)
    for (int tile = wave; tile < ntg; tile += 4) {
      f32x16 acc = tile_matmul(hs, ld, H, Wg, ldwg, tile * 32, 2 * H, lane);
      const int j = tile * 32 + ACC_COL(lane);
      if (j < 2 * H) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = ACC_ROW(r, lane);
          const int b = b0 + i;
          if (b >= B) continue;
          const int64_t row = (int64_t)b * T + t;
          float g = sigmoidf_(acc[r] + xproj[row * 3 * H + j]);
          gates[row * 3 * H + j] = g;
          if (j < H) rhs[i * ld + j] = g * hs[i * ld + j];
          else us[i * ld + (j - H)] = g;
        }
      }
    }
    __syncthreads();
    // c = tanh(xproj[:, 2H:3H] + (r*h).Wc);  h' = u*h + (1-u)*c
    for (int tile = wave; tile < ntc; tile += 4) {
      f32x16 acc = tile_matmul(rhs, ld, H, Wc, ldwc, tile * 32, H, lane);
      const int j = tile * 32 + ACC_COL(lane);
      if (j < H) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = ACC_ROW(r, lane);
          const int b = b0 + i;
          if (b >= B) continue;
          const int64_t row = (int64_t)b * T + t;
          float c = tanhf(acc[r] + xproj[row * 3 * H + 2 * H + j]);
          gates[row * 3 * H + 2 * H + j] = c;
          float u = us[i * ld + j], h = hs[i * ld + j];
          float hn = u * h + (1.0f - u) * c;
          bool live = t < length[b];
          out[row * ldo + j] = live ? hn : 0.f;   // dynamic_rnn: zero output past the length
          hs[i * ld + j] = live ? hn : h;         // ... and the state is carried through
        }
      }
    }
    __syncthreads();
  }
  if (final_state)
    for (int e = tid; e < RB * H; e += blockDim.x) {
      int i = e / H, j = e - i * H;
      if (b0 + i < B) final_state[(int64_t)(b0 + i) * H + j] = hs[i * ld + j];
    }
}

extern "C" int score_gru_fwd(int32_t B, int32_t T, int32_t H, const float* xproj, const float* Wg,
                             int32_t ldwg, const float* Wc, int32_t ldwc, const int32_t* length, float* out,
                             int32_t ldo, float* gates_save, float* final_state, void* stream) {
  if (!xproj || !Wg || !Wc || !length || !out || !gates_save || B <= 0 || T <= 0 || H <= 0) return SCORE_E_BADARG;
  if (gru_reg_ok(H)) {
    GruArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.T = T; a.H = H; a.length = length; a.nw8 = 1; a.x3_rec = 1;
    a.s[0].xproj = xproj; a.s[0].Wg = Wg; a.s[0].ldwg = ldwg; a.s[0].Wc = Wc; a.s[0].ldwc = ldwc;
    a.s[0].out = out; a.s[0].ldo = ldo; a.s[0].gates = gates_save; a.s[0].final_state = final_state;
    return score_gru_fwd_multi(a, 1, (hipStream_t)stream);
  }
  size_t lds = (size_t)3 * RB * (H + 1) * sizeof(float);
  if (lds > 160 * 1024) return SCORE_E_SHAPE;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)gru_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(gru_fwd_kernel, dim3((B + RB - 1) / RB), dim3(256), lds, (hipStream_t)stream, B, T, H, xproj,
                     Wg, ldwg, Wc, ldwc, length, out, ldo, gates_save, final_state);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// Backward recurrence.  WgT [2H][H] and WcT [H][H] are transposed copies of the
// recurrent weights (made by transpose_kernel below) so the MFMA B operand reads
// stay row-contiguous.
//   dh      = dout_t (live) + dh_next
//   du = dh*(h_prev - c), dc = dh*(1-u), dh_prev = dh*u
//   dpc = dc*(1-c^2);  d(rh) = dpc.Wc^T;  dr = d(rh)*h_prev;  dh_prev += d(rh)*r
//   dpr = dr*r*(1-r);  dpu = du*u*(1-u);  dh_prev += [dpr,dpu].Wg^T
__global__ __launch_bounds__(256) void gru_bwd_kernel(int B, int T, int H, const float* __restrict__ WgT,
                                                      const float* __restrict__ WcT,
                                                      const int32_t* __restrict__ length,
                                                      const float* __restrict__ out, int ldo,
                                                      const float* __restrict__ gates,
                                                      const float* __restrict__ dout, int lddo,
                                                      const float* __restrict__ dfinal,
                                                      float* __restrict__ dxproj, float* __restrict__ rh_out,
                                                      float* __restrict__ hprev_out) {
  extern __shared__ float sm[];
  const int ld = H + 1, ld2 = 2 * H + 1;
  float* dh = sm;                 // [RB][H+1]   running dL/dh
  float* dpc = dh + RB * ld;      // [RB][H+1]   candidate pre-activation grad
  float* dpg = dpc + RB * ld;     // [RB][2H+1]  gate pre-activation grads [dpr | dpu]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * RB;
  for (int e = tid; e < RB * H; e += blockDim.x) {
    int i = e / H, j = e - i * H;
    dh[i * ld + j] = (dfinal && b0 + i < B) ? dfinal[(int64_t)(b0 + i) * H + j] : 0.f;
  }
  __syncthreads();
  const int nth = (H + 31) / 32;
  for (int t = T - 1; t >= 0; --t) {
    // phase 1 (elementwise): dh_tot, du, dc -> dpu, dpc ; dh <- dh_tot*u
    for (int e = tid; e < RB * H; e += blockDim.x) {
      int i = e / H, j = e - i * H;
      int b = b0 + i;
      float v_dpc = 0.f, v_dpu = 0.f;
      if (b < B) {
        const int64_t row = (int64_t)b * T + t;
        const bool live = t < length[b];
        float hp = t > 0 ? out[(row - 1) * ldo + j] : 0.f;
        hprev_out[row * H + j] = live ? hp : 0.f;
        if (live) {
          float u = gates[row * 3 * H + H + j], c = gates[row * 3 * H + 2 * H + j];
          float d = dh[i * ld + j] + dout[row * lddo + j];
          float du = d * (hp - c), dc = d * (1.0f - u);
          v_dpu = du * u * (1.0f - u);
          v_dpc = dc * (1.0f - c * c);
          dh[i * ld + j] = d * u;
        }
        dxproj[row * 3 * H + H + j] = v_dpu;
        dxproj[row * 3 * H + 2 * H + j] = v_dpc;
      }
      dpc[i * ld + j] = v_dpc;
      dpg[i * ld2 + H + j] = v_dpu;
    }
    __syncthreads();
    // phase 2: d(rh) = dpc . Wc^T ; dr, dpr ; dh += d(rh)*r
    for (int tile = wave; tile < nth; tile += 4) {
      f32x16 acc = tile_matmul(dpc, ld, H, WcT, H, tile * 32, H, lane);
      const int j = tile * 32 + ACC_COL(lane);
      if (j < H) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = ACC_ROW(r, lane);
          const int b = b0 + i;
          float v_dpr = 0.f;
          if (b < B) {
            const int64_t row = (int64_t)b * T + t;
            const bool live = t < length[b];
            float rr = 0.f, hp = 0.f;
            if (live) {
              rr = gates[row * 3 * H + j];
              hp = t > 0 ? out[(row - 1) * ldo + j] : 0.f;
              float drh = acc[r];
              v_dpr = drh * hp * rr * (1.0f - rr);
              dh[i * ld + j] += drh * rr;
            }
            dxproj[row * 3 * H + j] = v_dpr;
            rh_out[row * H + j] = rr * hp;
          }
          dpg[i * ld2 + j] = v_dpr;
        }
      }
    }
    __syncthreads();
    // phase 3: dh += [dpr,dpu] . Wg^T
    for (int tile = wave; tile < nth; tile += 4) {
      f32x16 acc = tile_matmul(dpg, ld2, 2 * H, WgT, H, tile * 32, H, lane);
      const int j = tile * 32 + ACC_COL(lane);
      if (j < H) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = ACC_ROW(r, lane);
          dh[i * ld + j] += acc[r];
        }
      }
    }
    __syncthreads();
  }
}

__global__ void transpose_kernel(const float* __restrict__ src, int rows, int cols, int lds_, float* __restrict__ dst) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  int r = i / cols, c = i - r * cols;
  dst[(int64_t)c * rows + r] = src[(int64_t)r * lds_ + c];
}

// `hprev` must hold B*T*H + 3*H*H floats: the tail receives the transposed recurrent weights.
extern "C" int score_gru_bwd(int32_t B, int32_t T, int32_t H, const float* Wg, int32_t ldwg, const float* Wc,
                             int32_t ldwc, const int32_t* length, const float* out, int32_t ldo,
                             const float* gates_save, const float* dout, int32_t lddo, const float* dfinal,
                             float* dxproj, float* rh, float* hprev, void* stream) {
  if (!Wg || !Wc || !length || !out || !gates_save || !dout || !dxproj || !rh || !hprev || B <= 0 || T <= 0 ||
      H <= 0)
    return SCORE_E_BADARG;
  if (gru_reg_ok(H)) {
    GruArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.T = T; a.H = H; a.length = length; a.nw8 = 1; a.x3_rec = 1;
    GruSide& g = a.s[0];
    g.Wg = Wg; g.ldwg = ldwg; g.Wc = Wc; g.ldwc = ldwc; g.out = const_cast<float*>(out); g.ldo = ldo;
    g.gates = const_cast<float*>(gates_save); g.dout = dout; g.lddo = lddo; g.dfinal = dfinal;
    g.dxproj = dxproj; g.rh = rh; g.hprev = hprev;
    return score_gru_bwd_multi(a, 1, (hipStream_t)stream);
  }
  size_t lds = (size_t)RB * (2 * (H + 1) + (2 * H + 1)) * sizeof(float);
  if (lds > 160 * 1024) return SCORE_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  float* WgT = hprev + (int64_t)B * T * H;  // [2H][H]
  float* WcT = WgT + (int64_t)2 * H * H;    // [H][H]
  hipLaunchKernelGGL(transpose_kernel, dim3((2 * H * H + 255) / 256), dim3(256), 0, s, Wg, H, 2 * H, ldwg, WgT);
  SCORE_CHECK_LAUNCH();
  hipLaunchKernelGGL(transpose_kernel, dim3((H * H + 255) / 256), dim3(256), 0, s, Wc, H, H, ldwc, WcT);
  SCORE_CHECK_LAUNCH();
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)gru_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(gru_bwd_kernel, dim3((B + RB - 1) / RB), dim3(256), lds, s, B, T, H, WgT, WcT, length, out,
                     ldo, gates_save, dout, lddo, dfinal, dxproj, rh, hprev);
  SCORE_CHECK_LAUNCH();
  return 0;
}

// ======================================================================================
// Register-resident recurrence (H in {16,32,64,128}): the recurrent weights never leave
// the VGPRs.  A workgroup = 4 waves owns 16 batch rows; every wave owns a fixed set of
// 16-column output tiles and keeps their B operands for v_mfma_f32_16x16x4_f32 in registers
// for all T steps (H=128: 6 tiles x 32 k-steps = 192 VGPRs), so a step is LDS reads of the
// 16xH state, MFMAs and the fused pointwise epilogue -- no weight traffic at all.  Both GRUs
// of the model (user side / item side) share one launch.
// ======================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define RRB 16
// tools/gru_reg_probe.py builds this file with one ingredient stripped at a time (wrong results, timing only)
#if defined(GRP_NOMFMA)
#define GR_MFMA(a, b, c) ([&] { f32x4 t_ = (c); t_[0] += (a) * (b); return t_; }())
#else
#define GR_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif
#if defined(GRP_NOSTORE)
#define GR_STORE(lhs, v) do { if ((v) == 123.456f) lhs = (v); } while (0)
#else
#define GR_STORE(lhs, v) lhs = (v)
#endif
#if defined(GRP_NOXLOAD)
#define GR_XLOAD(e) 0.5f
#else
#define GR_XLOAD(e) (e)
#endif
// fast transcendental forms for the recurrence epilogues (v_exp_f32 / v_rcp_f32; abs error ~1e-7; __frcp_rn would be
// the ten-instruction correctly rounded division)
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }


// The time loops have no predicated memory operation at all (rows past the batch duplicate the last sample, see
// below).  That matters beyond the saved compares: with loads and stores under exec-mask branches the compiler's
// s_waitcnt placement falls back to vmcnt(0) in the loop, so every step waited for the prefetch it had just issued
// (43 % of the kernel's wave time in the first version).
template <int H, int NW>
__global__ __launch_bounds__(64 * NW) void gru_fwd_reg_kernel(const GruArgs a) {
  constexpr int KS = H / 4;                 // k-steps of 4
  constexpr int NTG = 2 * H / 16, NTC = H / 16;
  constexpr int TGW = (NTG + NW - 1) / NW, TCW = (NTC + NW - 1) / NW;
  constexpr bool TEX = (NTG % NW == 0) && (NTC % NW == 0);   // every wave owns whole tiles
  // K is dealt to the four lane quarters in contiguous runs (quarter lq owns k in [lq*KS, (lq+1)*KS)): a lane
  // then reads its A operands of four consecutive MFMA steps with one ds_read_b128, all of a phase's reads
  // go out before its MFMA chain starts.  Row stride H+4: 16-B aligned, the 16 rows of a read land on 64 banks.
  constexpr int LD = H + 4;
  __shared__ float hs[RRB * LD], rhs[RRB * LD], us[RRB * LD];
  const int tiles_b = (a.B + RRB - 1) / RRB;
  const int side = blockIdx.x / tiles_b;
  const GruSide& sd = a.s[side];
  const int b0 = (blockIdx.x - side * tiles_b) * RRB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, lq = lane >> 4;  // column inside a tile / k-quarter == output row group
  const int T = a.T;
  auto tile_ok = [&](int tile, int nt) { return TEX ? true : tile < nt; };

  float wg[TGW][KS], wc[TCW][KS];
#pragma unroll
  for (int tt = 0; tt < TGW; ++tt) {
    const int tile = wave + NW * tt;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      wg[tt][ks] = tile < NTG ? sd.Wg[(int64_t)(lq * KS + ks) * sd.ldwg + tile * 16 + lc] : 0.f;
  }
#pragma unroll
  for (int tt = 0; tt < TCW; ++tt) {
    const int tile = wave + NW * tt;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      wc[tt][ks] = tile < NTC ? sd.Wc[(int64_t)(lq * KS + ks) * sd.ldwc + tile * 16 + lc] : 0.f;
  }
  // Rows past the batch are DUPLICATES of the last sample: same inputs, same arithmetic, the same values stored to
  // the same addresses (a benign race) -- so a ragged batch needs no predicated memory operation either (with
  // stores under exec-mask branches the compiler's s_waitcnt placement falls back to vmcnt(0) in the time loop).
  int len[4];
  int64_t rowb[4];                // row of (sample, t = 0)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int bc = min(b0 + lq * 4 + r, a.B - 1);
    len[r] = a.length[bc];
    rowb[r] = (int64_t)bc * T;
  }
  constexpr bool rok[4] = {true, true, true, true};
  for (int e = tid; e < RRB * LD; e += 64 * NW) hs[e] = 0.f;
  __syncthreads();

  // x-projection values are read about one step ahead, unconditionally (clamped addresses): each half is fetched
  // again right after its last use (gate / candidate epilogue), into the registers it is consumed from -- a prefetch
  // into a second set is copied by v_mov behind a wait for the load just issued.  The first step is peeled: at the
  // loop header the waitcnt pass merges the prologue's state with the back edge's and keeps the stricter count.
  float xg[TGW][4], xc[TCW][4];
  auto fetch_xg = [&](int t) {
    const int tc = min(t, T - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* xr = sd.xproj + (rowb[r] + tc) * 3 * H;
#pragma unroll
      for (int tt = 0; tt < TGW; ++tt) xg[tt][r] = GR_XLOAD(xr[min(wave + NW * tt, NTG - 1) * 16 + lc]);
    }
  };
  auto fetch_xc = [&](int t) {
    const int tc = min(t, T - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* xr = sd.xproj + (rowb[r] + tc) * 3 * H;
#pragma unroll
      for (int tt = 0; tt < TCW; ++tt) xc[tt][r] = GR_XLOAD(xr[2 * H + min(wave + NW * tt, NTC - 1) * 16 + lc]);
    }
  };
  fetch_xg(0); fetch_xc(0);
  auto step = [&](const int t) {
    // gates = sigmoid(xproj[:, :2H] + h . Wg)
    f32x4 acc[TGW];
#pragma unroll
    for (int tt = 0; tt < TGW; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      float4 av4[KS / 4];
#pragma unroll
      for (int q = 0; q < KS / 4; ++q) av4[q] = *reinterpret_cast<const float4*>(&hs[lc * LD + lq * KS + 4 * q]);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float av = (ks & 3) == 0 ? av4[ks >> 2].x : (ks & 3) == 1 ? av4[ks >> 2].y : (ks & 3) == 2 ? av4[ks >> 2].z : av4[ks >> 2].w;
#pragma unroll
        for (int tt = 0; tt < TGW; ++tt) acc[tt] = GR_MFMA(av, wg[tt][ks], acc[tt]);
      }
    }
#pragma unroll
    for (int tt = 0; tt < TGW; ++tt) {
      const int tile = wave + NW * tt;
      if (!tile_ok(tile, NTG)) continue;
      const int j = tile * 16 + lc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = lq * 4 + r;
        const float g = sigmoid_fast(acc[tt][r] + xg[tt][r]);
        if (rok[r]) GR_STORE(sd.gates[(rowb[r] + t) * 3 * H + j], g);
        if (j < H) rhs[i * LD + j] = g * hs[i * LD + j];
        else us[i * LD + (j - H)] = g;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    fetch_xg(t + 1);
    __syncthreads();
    // c = tanh(xproj[:, 2H:] + (r*h) . Wc) ; h' = u*h + (1-u)*c
    f32x4 acc2[TCW];
#pragma unroll
    for (int tt = 0; tt < TCW; ++tt) acc2[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      float4 av4[KS / 4];
#pragma unroll
      for (int q = 0; q < KS / 4; ++q) av4[q] = *reinterpret_cast<const float4*>(&rhs[lc * LD + lq * KS + 4 * q]);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float av = (ks & 3) == 0 ? av4[ks >> 2].x : (ks & 3) == 1 ? av4[ks >> 2].y : (ks & 3) == 2 ? av4[ks >> 2].z : av4[ks >> 2].w;
#pragma unroll
        for (int tt = 0; tt < TCW; ++tt)
          acc2[tt] = GR_MFMA(av, wc[tt][ks], acc2[tt]);
      }
    }
#pragma unroll
    for (int tt = 0; tt < TCW; ++tt) {
      const int tile = wave + NW * tt;
      if (!tile_ok(tile, NTC)) continue;
      const int j = tile * 16 + lc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = lq * 4 + r;
        const float c = tanh_fast(acc2[tt][r] + xc[tt][r]);
        const float u = us[i * LD + j], h = hs[i * LD + j];
        const float hn = u * h + (1.0f - u) * c;
        const bool live = t < len[r];
        if (rok[r]) {
          const int64_t row = rowb[r] + t;
          GR_STORE(sd.gates[row * 3 * H + 2 * H + j], c);
          GR_STORE(sd.out[row * sd.ldo + j], (live ? hn : 0.f));
        }
        hs[i * LD + j] = live ? hn : h;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    fetch_xc(t + 1);
    __syncthreads();
  };
  step(0);
  for (int t = 1; t < T; ++t) step(t);
  if (sd.final_state)
    for (int e = tid; e < RRB * H; e += 64 * NW) {
      const int i = e / H, j = e - i * H;
      if (b0 + i < a.B) sd.final_state[(int64_t)(b0 + i) * H + j] = hs[i * LD + j];
    }
}

template <int H, int NW>
__global__ __launch_bounds__(64 * NW) void gru_bwd_reg_kernel(const GruArgs a) {
  constexpr int KS = H / 4;
  constexpr int NT = H / 16;
  constexpr int TW = (NT + NW - 1) / NW;
  constexpr bool TEX = NT % NW == 0;
  constexpr int LD = H + 4, LD2 = 2 * H + 4;     // K dealt to the lane quarters in contiguous runs, as in the forward
  __shared__ float dh[RRB * LD], dpc[RRB * LD], dpg[RRB * LD2];
  const int tiles_b = (a.B + RRB - 1) / RRB;
  const int side = blockIdx.x / tiles_b;
  const GruSide& sd = a.s[side];
  const int b0 = (blockIdx.x - side * tiles_b) * RRB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, lq = lane >> 4;
  const int T = a.T;
  auto tile_ok = [&](int tile) { return TEX ? true : tile < NT; };

  // B operands of the two transposed products: B[k][j] = Wc[j][k] (k < H), Wg[j][k] (k < 2H)
  float wct[TW][KS], wgt[TW][2 * KS];
#pragma unroll
  for (int tt = 0; tt < TW; ++tt) {
    const int tile = wave + NW * tt;
    const int j = tile * 16 + lc;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wct[tt][ks] = tile < NT ? sd.Wc[(int64_t)j * sd.ldwc + lq * KS + ks] : 0.f;
#pragma unroll
    for (int ks = 0; ks < 2 * KS; ++ks) wgt[tt][ks] = tile < NT ? sd.Wg[(int64_t)j * sd.ldwg + lq * 2 * KS + ks] : 0.f;
  }
  // every thread owns the elements (row i = lq*4 + r, column j = (wave + NW*tt)*16 + lc) in all three
  // phases, so the saved activations of a step are read once, about one step ahead of their use.
  // Rows past the batch are duplicates of the last sample (see the forward): no predicated memory operation.
  int len[4], bcs[4];
  int64_t rowb[4];
  constexpr bool rok[4] = {true, true, true, true};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    bcs[r] = min(b0 + lq * 4 + r, a.B - 1);
    len[r] = a.length[bcs[r]];
    rowb[r] = (int64_t)bcs[r] * T;
  }
#pragma unroll
  for (int tt = 0; tt < TW; ++tt) {
    const int tile = wave + NW * tt;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = lq * 4 + r, j = tile * 16 + lc;
      if (tile < NT) dh[i * LD + j] = sd.dfinal ? sd.dfinal[(int64_t)bcs[r] * H + j] : 0.f;
    }
  }
  // Saved activations, read unconditionally (clamped addresses): each array is fetched again right after its last
  // use in the step (u, c, dout: phase 1; r: phase 2), into the registers it is consumed from; only h_prev, which
  // both phases and the stores need masked, keeps a copy.  (A prefetch into a second register set is copied by
  // v_mov behind a wait for the load just issued; the first step is peeled for the loop header's waitcnt state.)
  float n_u[TW][4], n_c[TW][4], n_r[TW][4], n_hp[TW][4], n_do[TW][4];
  auto fetch_ucd = [&](int t) {
    const int tc = max(t, 0);
#pragma unroll
    for (int tt = 0; tt < TW; ++tt) {
      const int j = min(wave + NW * tt, NT - 1) * 16 + lc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = rowb[r] + tc;
        n_u[tt][r] = GR_XLOAD(sd.gates[row * 3 * H + H + j]);
        n_c[tt][r] = GR_XLOAD(sd.gates[row * 3 * H + 2 * H + j]);
        n_do[tt][r] = GR_XLOAD(sd.dout[row * sd.lddo + j]);
      }
    }
  };
  auto fetch_r = [&](int t) {
    const int tc = max(t, 0);
#pragma unroll
    for (int tt = 0; tt < TW; ++tt) {
      const int j = min(wave + NW * tt, NT - 1) * 16 + lc;
#pragma unroll
      for (int r = 0; r < 4; ++r) n_r[tt][r] = GR_XLOAD(sd.gates[(rowb[r] + tc) * 3 * H + j]);
    }
  };
  auto fetch_hp = [&](int t) {
    const int tp = max(t - 1, 0);          // (h_prev of t = 0 is never used)
#pragma unroll
    for (int tt = 0; tt < TW; ++tt) {
      const int j = min(wave + NW * tt, NT - 1) * 16 + lc;
#pragma unroll
      for (int r = 0; r < 4; ++r) n_hp[tt][r] = GR_XLOAD(sd.out[(rowb[r] + tp) * sd.ldo + j]);
    }
  };
  fetch_ucd(T - 1); fetch_r(T - 1); fetch_hp(T - 1);
  __syncthreads();

  auto step = [&](const int t) {
    float c_hp[TW][4];                  // h_{t-1}, 0 past the length and at t = 0
#pragma unroll
    for (int tt = 0; tt < TW; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) c_hp[tt][r] = (t < len[r] && t > 0) ? n_hp[tt][r] : 0.f;
    __builtin_amdgcn_sched_barrier(0);
    fetch_hp(t - 1);
    // phase 1 (elementwise): dpu, dpc ; dh <- dh_tot * u      (selects, no branches)
#pragma unroll
    for (int tt = 0; tt < TW; ++tt) {
      const int tile = wave + NW * tt;
      if (!tile_ok(tile)) continue;
      const int j = tile * 16 + lc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = lq * 4 + r;
        const bool live = t < len[r];
        const float u = n_u[tt][r], c = n_c[tt][r];      // (a dead step's results are discarded by `live`)
        const float dold = dh[i * LD + j];
        const float d = dold + n_do[tt][r];
        const float du = d * (c_hp[tt][r] - c), dc = d * (1.0f - u);
        const float v_dpu = live ? du * u * (1.0f - u) : 0.f;
        const float v_dpc = live ? dc * (1.0f - c * c) : 0.f;
        dh[i * LD + j] = live ? d * u : dold;
        if (rok[r]) {
          const int64_t row = rowb[r] + t;
          GR_STORE(sd.hprev[row * H + j], c_hp[tt][r]);
          GR_STORE(sd.dxproj[row * 3 * H + H + j], v_dpu);
          GR_STORE(sd.dxproj[row * 3 * H + 2 * H + j], v_dpc);
        }
        dpc[i * LD + j] = v_dpc;
        dpg[i * LD2 + H + j] = v_dpu;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    fetch_ucd(t - 1);
    __syncthreads();
    // phase 2: d(rh) = dpc . Wc^T ; dpr = d(rh)*h_prev*r(1-r) ; dh += d(rh)*r
    {
      f32x4 acc[TW];
#pragma unroll
      for (int tt = 0; tt < TW; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      float4 av4[KS / 4];
#pragma unroll
      for (int q = 0; q < KS / 4; ++q) av4[q] = *reinterpret_cast<const float4*>(&dpc[lc * LD + lq * KS + 4 * q]);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float av = (ks & 3) == 0 ? av4[ks >> 2].x : (ks & 3) == 1 ? av4[ks >> 2].y : (ks & 3) == 2 ? av4[ks >> 2].z : av4[ks >> 2].w;
#pragma unroll
        for (int tt = 0; tt < TW; ++tt)
          acc[tt] = GR_MFMA(av, wct[tt][ks], acc[tt]);
      }
#pragma unroll
      for (int tt = 0; tt < TW; ++tt) {
        const int tile = wave + NW * tt;
        if (!tile_ok(tile)) continue;
        const int j = tile * 16 + lc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = lq * 4 + r;
          const bool live = t < len[r];
          const float rr = live ? n_r[tt][r] : 0.f, hp = c_hp[tt][r];   // both 0 past the length
          const float drh = acc[tt][r];
          const float v_dpr = live ? drh * hp * rr * (1.0f - rr) : 0.f;
          dh[i * LD + j] += live ? drh * rr : 0.f;
          if (rok[r]) {
            const int64_t row = rowb[r] + t;
            GR_STORE(sd.dxproj[row * 3 * H + j], v_dpr);
            GR_STORE(sd.rh[row * H + j], (rr * hp));
          }
          dpg[i * LD2 + j] = v_dpr;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      fetch_r(t - 1);
    }
    __syncthreads();
    // phase 3: dh += [dpr | dpu] . Wg^T
    {
      f32x4 acc[TW];
#pragma unroll
      for (int tt = 0; tt < TW; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      float4 av4[KS / 2];
#pragma unroll
      for (int q = 0; q < KS / 2; ++q) av4[q] = *reinterpret_cast<const float4*>(&dpg[lc * LD2 + lq * 2 * KS + 4 * q]);
#pragma unroll
      for (int ks = 0; ks < 2 * KS; ++ks) {
        const float av = (ks & 3) == 0 ? av4[ks >> 2].x : (ks & 3) == 1 ? av4[ks >> 2].y : (ks & 3) == 2 ? av4[ks >> 2].z : av4[ks >> 2].w;
#pragma unroll
        for (int tt = 0; tt < TW; ++tt)
          acc[tt] = GR_MFMA(av, wgt[tt][ks], acc[tt]);
      }
#pragma unroll
      for (int tt = 0; tt < TW; ++tt) {
        const int tile = wave + NW * tt;
        if (!tile_ok(tile)) continue;
        const int j = tile * 16 + lc;
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[(lq * 4 + r) * LD + j] += acc[tt][r];
      }
    }
    __syncthreads();
  };
  step(T - 1);
  for (int t = T - 2; t >= 0; --t) step(t);
}

// ------------------------------------------------------------------ step-by-step recurrence (any H)
// Per time slice: gpre = h.Wg (grouped GEMM over the sides) -> gates, r*h -> cpre = (r*h).Wc -> candidate,
// new state.  The same arithmetic as gru_fwd_kernel / gru_bwd_kernel, with the matrix products on the GEMM
// kernels (bf16x3 where allowed) instead of one-float-per-lane operand loads.
struct GruStepSide {
  const float* xproj; float* out; int ldo; float* gates; float* hstate; const float* gpre; float* rh;
  const float* cpre;
  // backward
  const float* dout; int lddo; float* dxproj; float* rh_out; float* hprev_out; float* dh; float* dpc;
  const float* drh; float* dpg;
};
struct GruStepArgs { GruStepSide s[2]; const int32_t* length; int B, T, H, t; };

__global__ void gru_step_gates_kernel(const GruStepArgs a, int first) {
  const GruStepSide& sd = a.s[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)a.B * 2 * a.H) return;
  const int b = (int)(i / (2 * a.H)), j = (int)(i - (int64_t)b * 2 * a.H);
  const int64_t row = (int64_t)b * a.T + a.t;
  const float pre = first ? 0.f : sd.gpre[i];
  const float g = sigmoidf_(pre + sd.xproj[row * 3 * a.H + j]);
  sd.gates[row * 3 * a.H + j] = g;
  if (j < a.H) sd.rh[(int64_t)b * a.H + j] = first ? 0.f : g * sd.hstate[(int64_t)b * a.H + j];
}
__global__ void gru_step_out_kernel(const GruStepArgs a, int first) {
  const GruStepSide& sd = a.s[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)a.B * a.H) return;
  const int b = (int)(i / a.H), j = (int)(i - (int64_t)b * a.H);
  const int64_t row = (int64_t)b * a.T + a.t;
  const float pre = first ? 0.f : sd.cpre[i];
  const float c = tanhf(pre + sd.xproj[row * 3 * a.H + 2 * a.H + j]);
  sd.gates[row * 3 * a.H + 2 * a.H + j] = c;
  const float u = sd.gates[row * 3 * a.H + a.H + j], h = first ? 0.f : sd.hstate[i];
  const float hn = u * h + (1.0f - u) * c;
  const bool live = a.t < a.length[b];
  sd.out[row * sd.ldo + j] = live ? hn : 0.f;     // dynamic_rnn: zero output past the length
  sd.hstate[i] = live ? hn : h;                   // ... and the state is carried through
}
// backward phase 1: dh_tot, du, dc -> dpu, dpc ; dh <- dh_tot*u
__global__ void gru_bstep_a_kernel(const GruStepArgs a) {
  const GruStepSide& sd = a.s[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)a.B * a.H) return;
  const int b = (int)(i / a.H), j = (int)(i - (int64_t)b * a.H), H = a.H, t = a.t;
  const int64_t row = (int64_t)b * a.T + t;
  const bool live = t < a.length[b];
  const float hp = t > 0 ? sd.out[(row - 1) * sd.ldo + j] : 0.f;
  sd.hprev_out[row * H + j] = live ? hp : 0.f;
  float v_dpc = 0.f, v_dpu = 0.f;
  if (live) {
    const float u = sd.gates[row * 3 * H + H + j], c = sd.gates[row * 3 * H + 2 * H + j];
    const float d = sd.dh[i] + sd.dout[row * sd.lddo + j];
    const float du = d * (hp - c), dc = d * (1.0f - u);
    v_dpu = du * u * (1.0f - u);
    v_dpc = dc * (1.0f - c * c);
    sd.dh[i] = d * u;
  }
  sd.dxproj[row * 3 * H + H + j] = v_dpu;
  sd.dxproj[row * 3 * H + 2 * H + j] = v_dpc;
  sd.dpc[i] = v_dpc;
  sd.dpg[(int64_t)b * 2 * H + H + j] = v_dpu;
}
// backward phase 2 (after drh = dpc . Wc^T): dr, dpr ; dh += d(rh)*r
__global__ void gru_bstep_b_kernel(const GruStepArgs a) {
  const GruStepSide& sd = a.s[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)a.B * a.H) return;
  const int b = (int)(i / a.H), j = (int)(i - (int64_t)b * a.H), H = a.H, t = a.t;
  const int64_t row = (int64_t)b * a.T + t;
  const bool live = t < a.length[b];
  float v_dpr = 0.f, rr = 0.f, hp = 0.f;
  if (live) {
    rr = sd.gates[row * 3 * H + j];
    hp = t > 0 ? sd.out[(row - 1) * sd.ldo + j] : 0.f;
    const float drh = sd.drh[i];
    v_dpr = drh * hp * rr * (1.0f - rr);
    sd.dh[i] += drh * rr;
  }
  sd.dxproj[row * 3 * H + j] = v_dpr;
  sd.rh_out[row * H + j] = rr * hp;
  sd.dpg[(int64_t)b * 2 * H + j] = v_dpr;
}
__global__ void gru_copy_or_zero_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src ? src[i] : 0.f;
}

static int gru_fwd_steps(GruArgs& a, int nsides, hipStream_t s) {
  const int B = a.B, T = a.T, H = a.H;
  const int64_t BH = (int64_t)B * H;
  if (a.tmp_floats < 10 * BH) return SCORE_E_WORKSPACE;
  GruStepArgs g;
  memset(&g, 0, sizeof(g));
  g.length = a.length; g.B = B; g.T = T; g.H = H;
  const float *Ah[2], *Wg[2], *Arh[2], *Wc[2];
  float *Cg[2], *Cc[2];
  for (int i = 0; i < nsides; ++i) {
    float* base = a.tmp + (int64_t)i * 5 * BH;
    GruStepSide& sd = g.s[i];
    sd.xproj = a.s[i].xproj; sd.out = a.s[i].out; sd.ldo = a.s[i].ldo; sd.gates = a.s[i].gates;
    sd.hstate = base; sd.gpre = base + BH; sd.rh = base + 3 * BH; sd.cpre = base + 4 * BH;
    Ah[i] = sd.hstate; Wg[i] = a.s[i].Wg; Cg[i] = base + BH; Arh[i] = sd.rh; Wc[i] = a.s[i].Wc; Cc[i] = base + 4 * BH;
  }
  const unsigned g2 = (unsigned)cdiv64(2 * BH, 256), g1 = (unsigned)cdiv64(BH, 256);
  for (int t = 0; t < T; ++t) {
    g.t = t;
    const int first = t == 0;      // h_{-1} = 0: both products vanish
    if (!first)
      SCORE_TRY(score_gemm_same_shape(0, nsides, B, 2 * H, H, Ah, H, Wg, a.s[0].ldwg, Cg, 2 * H, 0, a.x3, nullptr, 0, s));
    hipLaunchKernelGGL(gru_step_gates_kernel, dim3(g2, nsides), dim3(256), 0, s, g, first);
    SCORE_CHECK_LAUNCH();
    if (!first)
      SCORE_TRY(score_gemm_same_shape(0, nsides, B, H, H, Arh, H, Wc, a.s[0].ldwc, Cc, H, 0, a.x3, nullptr, 0, s));
    hipLaunchKernelGGL(gru_step_out_kernel, dim3(g1, nsides), dim3(256), 0, s, g, first);
    SCORE_CHECK_LAUNCH();
  }
  for (int i = 0; i < nsides; ++i)
    if (a.s[i].final_state) {
      hipLaunchKernelGGL(gru_copy_or_zero_kernel, dim3(g1), dim3(256), 0, s, a.s[i].final_state, g.s[i].hstate, BH);
      SCORE_CHECK_LAUNCH();
    }
  return 0;
}

static int gru_bwd_steps(GruArgs& a, int nsides, hipStream_t s) {
  const int B = a.B, T = a.T, H = a.H;
  const int64_t BH = (int64_t)B * H;
  if (a.tmp_floats < 10 * BH) return SCORE_E_WORKSPACE;
  GruStepArgs g;
  memset(&g, 0, sizeof(g));
  g.length = a.length; g.B = B; g.T = T; g.H = H;
  const float *Adpc[2], *Wc[2], *Adpg[2], *Wg[2];
  float *Cdrh[2], *Cdh[2];
  const unsigned g1 = (unsigned)cdiv64(BH, 256);
  for (int i = 0; i < nsides; ++i) {
    float* base = a.tmp + (int64_t)i * 5 * BH;
    GruStepSide& sd = g.s[i];
    sd.out = a.s[i].out; sd.ldo = a.s[i].ldo; sd.gates = a.s[i].gates; sd.dout = a.s[i].dout; sd.lddo = a.s[i].lddo;
    sd.dxproj = a.s[i].dxproj; sd.rh_out = a.s[i].rh; sd.hprev_out = a.s[i].hprev;
    sd.dh = base; sd.dpc = base + BH; sd.drh = base + 2 * BH; sd.dpg = base + 3 * BH;
    Adpc[i] = sd.dpc; Wc[i] = a.s[i].Wc; Cdrh[i] = base + 2 * BH; Adpg[i] = sd.dpg; Wg[i] = a.s[i].Wg; Cdh[i] = sd.dh;
    hipLaunchKernelGGL(gru_copy_or_zero_kernel, dim3(g1), dim3(256), 0, s, sd.dh, a.s[i].dfinal, BH);
    SCORE_CHECK_LAUNCH();
  }
  for (int t = T - 1; t >= 0; --t) {
    g.t = t;
    hipLaunchKernelGGL(gru_bstep_a_kernel, dim3(g1, nsides), dim3(256), 0, s, g);
    SCORE_CHECK_LAUNCH();
    // d(rh) = dpc . Wc^T   (Wc is [H_in, H_out]: the NT layout reads it as is)
    SCORE_TRY(score_gemm_same_shape(1, nsides, B, H, H, Adpc, H, Wc, a.s[0].ldwc, Cdrh, H, 0, a.x3, nullptr, 0, s));
    hipLaunchKernelGGL(gru_bstep_b_kernel, dim3(g1, nsides), dim3(256), 0, s, g);
    SCORE_CHECK_LAUNCH();
    // dh += [dpr | dpu] . Wg^T
    SCORE_TRY(score_gemm_same_shape(1, nsides, B, H, 2 * H, Adpg, 2 * H, Wg, a.s[0].ldwg, Cdh, H, 4, a.x3, nullptr, 0, s));
  }
  return 0;
}

static bool gru_reg_ok(int H) { return H == 16 || H == 32 || H == 64 || H == 128; }
// (the H = 128 recurrences on the f32-input MFMA instead of the bf16x3 form: score_state_t.debug_flags bit 2 -> GruArgs.x3_rec)
static bool gru_x3_allowed() { return true; }

int score_gru_fwd_multi(GruArgs& a, int nsides, hipStream_t s) {
  const int H = a.H;
  if (a.x3_rec && gru_x3_allowed() && score_gru_x3_ok(H, a.nw8)) return score_gru_fwd_x3(a, nsides, s);
  if (gru_reg_ok(H)) {
    dim3 grid(nsides * ((a.B + RRB - 1) / RRB));
#define LF(Hv, NWv) hipLaunchKernelGGL((gru_fwd_reg_kernel<Hv, NWv>), grid, dim3(64 * NWv), 0, s, a)
    if (H == 16) LF(16, 4);
    else if (H == 32) LF(32, 4);
    else if (H == 64) LF(64, 4);
    else if (a.nw8) LF(128, 8);
    else LF(128, 4);
#undef LF
    SCORE_CHECK_LAUNCH();
    return 0;
  }
  if (!a.stepwise && score_gru_stream_ok(H) && a.tmp && a.tmp_floats >= score_gru_stream_tmp_floats(H, nsides))
    return score_gru_fwd_stream(a, nsides, s);
  if (a.tmp && a.tmp_floats >= 10 * (int64_t)a.B * H && (nsides == 1 || (a.s[0].ldwg == a.s[1].ldwg && a.s[0].ldwc == a.s[1].ldwc)))
    return gru_fwd_steps(a, nsides, s);
  for (int i = 0; i < nsides; ++i) {
    const GruSide& sd = a.s[i];
    SCORE_TRY(score_gru_fwd(a.B, a.T, H, sd.xproj, sd.Wg, sd.ldwg, sd.Wc, sd.ldwc, a.length, sd.out, sd.ldo,
                            sd.gates, sd.final_state, s));
  }
  return 0;
}

int score_gru_bwd_multi(GruArgs& a, int nsides, hipStream_t s) {
  const int H = a.H;
  if (a.x3_rec && gru_x3_allowed() && score_gru_x3_ok(H, a.nw8)) return score_gru_bwd_x3(a, nsides, s);
  if (gru_reg_ok(H)) {
    dim3 grid(nsides * ((a.B + RRB - 1) / RRB));
#define LB(Hv, NWv) hipLaunchKernelGGL((gru_bwd_reg_kernel<Hv, NWv>), grid, dim3(64 * NWv), 0, s, a)
    if (H == 16) LB(16, 4);
    else if (H == 32) {     // two waves from 20 slices on: a shorter step, a longer prologue (each wave holds twice the weights) -- the CCMR
      if (a.T >= 20) LB(32, 2);     // shape (T = 40) 0.5174 -> 0.5064 ms/step, cfg-2 (T = 10) 0.272 -> 0.278; the forward keeps four
      else LB(32, 4);
    }
    else if (H == 64) LB(64, 4);
    else if (a.nw8) LB(128, 8);
    else LB(128, 4);
#undef LB
    SCORE_CHECK_LAUNCH();
    return 0;
  }
  if (!a.stepwise && score_gru_stream_ok(H) && a.tmp && a.tmp_floats >= score_gru_stream_tmp_floats(H, nsides))
    return score_gru_bwd_stream(a, nsides, s);
  if (a.tmp && a.tmp_floats >= 10 * (int64_t)a.B * H && (nsides == 1 || (a.s[0].ldwg == a.s[1].ldwg && a.s[0].ldwc == a.s[1].ldwc)))
    return gru_bwd_steps(a, nsides, s);
  for (int i = 0; i < nsides; ++i) {
    const GruSide& sd = a.s[i];
    SCORE_TRY(score_gru_bwd(a.B, a.T, H, sd.Wg, sd.ldwg, sd.Wc, sd.ldwc, a.length, sd.out, sd.ldo, sd.gates,
                            sd.dout, sd.lddo, sd.dfinal, sd.dxproj, sd.rh, sd.hprev, s));
  }
  return 0;
}
