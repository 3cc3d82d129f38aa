// The LSTM recurrence of questionNet as sequence-level entry points: vqa_lstm_seq_fwd / vqa_lstm_seq_bwd.
//
// Reference: nn.LSTM inside questionNet (models/model.py:145-149, 159-164): per step and direction
//   [i f g o] = x_t W_ih^T + b_ih + h_{t-1} W_hh^T + b_hh;  c' = s(f) c + s(i) tanh(g);  h' = s(o) tanh(c'),
// with the packed-sequence rule that sample b only advances while t < q_len[b]; the reverse direction visits
// t = T-1 .. 0.  The x half is one big GEMM over all T (vqa_gemm, xg [T*B][4H]); the recurrent half is here.
//
// One launch per time step covers BOTH directions (2 x 256 workgroups at B = 256, H = 1024: every CU busy), and a
// whole sequence is ONE C-ABI call: T dependent launches, either enqueued one by one or replayed as an explicit
// hipGraph that is built once per distinct argument set and cached (the training loop presents the same buffers
// every step, so steady state is one hipGraphLaunch per direction-pair chain).
//
//   forward step   GEMM engine of gemm_core.hpp, 64x64 tiles: A = h_{t-1} [B][H] (type R), B = W_hh rows GATHERED so
//                  that a workgroup's 64 output columns are the four gates of 16 hidden units; epilogue = the cell
//                  (x-projection added, own activation per lane, the 4 lanes of a unit exchange gates), writing
//                  c', h', the saved gate activations and, on the last step, the final cell state.
//   backward step  dh_{prev} = dgates_t . W_hh is a [B x 4H] . [4H x H] product: small output, long K.  A workgroup
//                  owns a 64 x 32 output tile over the WHOLE K = 4H, split over its 8 MFMA waves by gate slice
//                  (wave = (row half, gate)), partial tiles combined through LDS in a fixed order -- no split-K slabs
//                  and no reduce launch -- and the epilogue is the cell backward of the NEXT step to be
//                  differentiated: it turns dh_{prev} (+ the pass-through of finished samples) straight into that
//                  step's dgates, which is the A operand of the next launch.  One launch per step replaces
//                  lstm_cell_bwd + split-K GEMM + splitk_reduce.
#include <array>
#include <map>
#include <mutex>

#include "gemm_core.hpp"

namespace vqa {

// ------------------------------------------------------------------ direction descriptors (device view)
struct SeqDir {
  const float* w_hh;   // [4H][H]
  const float* xg;     // [T][B][4H]           forward: x-projection + both biases
  float* gates;        // [T][B][4H]           saved gate activations (forward writes, backward reads)
  float* Hs;           // [T+1][B][H]          state chains; forward direction: slot t -> t+1, reverse: t+1 -> t
  float* Cs;
  float* c_final;      // [B][cf_ld] or null   final cell state (forward)
  float* dgates;       // [T][B][4H]           backward: pre-activation gradients, out
  float* dh;           // [B][H]               backward work: gradient w.r.t. h (in place)
  float* dc;           // [B][H]               backward work: gradient w.r.t. c (in place)
  int reverse;
};
struct SeqArgs {
  SeqDir d[2];
  const int64_t* q_len;
  int64_t cf_ld;
  int ndir, B, T, H;
};

// time visited at forward step n / at backward (BPTT) step n, and the state slots of a time
__device__ __forceinline__ int fwd_time(int reverse, int n, int T) { return reverse ? T - 1 - n : n; }
__device__ __forceinline__ int bwd_time(int reverse, int n, int T) { return reverse ? n : T - 1 - n; }
__device__ __forceinline__ int slot_in(int reverse, int t) { return reverse ? t + 1 : t; }
__device__ __forceinline__ int slot_out(int reverse, int t) { return reverse ? t : t + 1; }

// Activations of the fused epilogues: one v_exp_f32 + one v_rcp_f32 each (about 1 ulp apiece; |error| <= 3e-7 on
// values in [-1, 1]), the same code for sigmoid and tanh lanes, so a wave whose lanes hold different gates does not
// execute two library routines under predication (the 16 elements of a lane cost ~100 VALU instructions less each).
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) { return 2.0f * fast_sigmoid(2.0f * x) - 1.0f; }

// ------------------------------------------------------------------ forward step
// Type R loader over W_hh [4H][H] whose tile rows are gathered gate-major per wave column block.
template <int NV, int LT = 256>
struct LstmWhhR {
  struct Params { const float* p; int H; };
  struct Raw { float4 v[NV]; };
  static constexpr bool kTypeR = true;
  const float* base;
  uint32_t voff[NV];
  int K;
  __device__ __forceinline__ void init(const Params& q, int n0, int tid, int /*ks0*/) {
    K = q.H;
    base = q.p;
    const int c4 = 4 * StageMap<LT>::r_chunk(tid);
    const int u0 = n0 / 4;                                   // 64 columns = 16 units x 4 gates
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int rl = StageMap<LT>::r_row(tid, p);            // tile column 0..63
      const int wn = rl >> 5, cc = rl & 31;
      const int unit = u0 + 8 * wn + (cc & 7), gate = cc >> 3;
      voff[p] = unit < q.H ? (uint32_t)((gate * q.H + unit) * q.H + c4) * 4u : BUF_OOB;
    }
  }
  __device__ __forceinline__ void issue(int ks, Raw& r) {
    // H % BK == 0 (checked on entry): no K tail; past the end the resource is empty and every lane reads zeros
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(base + ks * BK, ks * BK < K ? BUF_OOB : 0u);
#pragma unroll
    for (int p = 0; p < NV; ++p) r.v[p] = buf_load16(rs, voff[p]);
  }
  __device__ __forceinline__ void finish(const Raw& r, float4 (&o)[NV]) const {
#pragma unroll
    for (int p = 0; p < NV; ++p) o[p] = r.v[p];
  }
};

using CfgL = TileCfg<64, 64, 2, 2>;

// grid = ndir * tiles_m * tiles_n; logical ids (after the XCD swizzle) run direction, column tile, row tile: an XCD
// keeps a contiguous slice of ONE direction's W_hh (weight-stationary order).
__global__ __launch_bounds__(CfgL::THREADS, CfgL::MIN_WAVES) void lstm_step_fwd_kernel(SeqArgs a, int n, int tiles_m,
                                                                                    int tiles_n) {
  using Cfg = CfgL;
  using AL = PlainR<Cfg::NVA, Cfg::LT>;
  using BL = LstmWhhR<Cfg::NVB, Cfg::LT>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const int B = a.B, T = a.T, H = a.H;
  const int tiles = tiles_m * tiles_n;
  const int lid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int dir = lid / tiles, u = lid - dir * tiles;
  const int nt = u / tiles_m, mt = u - nt * tiles_m;
  const SeqDir D = dir ? a.d[1] : a.d[0];
  const int t = fwd_time(D.reverse, n, T);
  const int64_t BH = (int64_t)B * H;
  const float* h_in = D.Hs + slot_in(D.reverse, t) * BH;
  const float* c_in = D.Cs + slot_in(D.reverse, t) * BH;
  float* h_out = D.Hs + slot_out(D.reverse, t) * BH;
  float* c_out = D.Cs + slot_out(D.reverse, t) * BH;
  const float* xg = D.xg + (int64_t)t * B * 4 * H;
  float* gates = D.gates + (int64_t)t * B * 4 * H;
  float* c_final = (n == T - 1) ? D.c_final : nullptr;
  const int m0 = mt * Cfg::BM, n0 = nt * Cfg::BN;
  f32x16 acc[Cfg::TM][Cfg::TN];
  acc_zero<Cfg>(acc);
  if (!gemm_mainloop<Cfg, AL, BL>(
          [&](AL& al, BL& bl) {
            al.init(typename AL::Params{h_in, (int64_t)H, B, H}, m0, loader_tid<Cfg>(), 0);
            bl.init(typename BL::Params{D.w_hh, H}, n0, loader_tid<Cfg>(), 0);
          },
          [](AL&, BL&) {}, acc, 0, H / BK, H, smem))
    return;
  // ---- cell epilogue: loads first (one batch), then the per-element code (stores only)
  const int l31 = lane & 31, hh = lane >> 5;
  const int gate = l31 >> 3, unit = n0 / 4 + 8 * wn + (l31 & 7);
  const int row0 = m0 + 32 * wm + 4 * hh;
  const bool uok = unit < H;
  float xv[16], cp[16], hp[16];
  bool act[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + (r & 3) + 8 * (r >> 2);
    const int rr = row < B ? row : 0, uu = uok ? unit : 0;
    xv[r] = xg[(int64_t)rr * 4 * H + gate * H + uu];
    cp[r] = c_in[(int64_t)rr * H + uu];
    hp[r] = h_in[(int64_t)rr * H + uu];
    act[r] = (int64_t)t < a.q_len[rr];
  }
  const int src = lane & ~24;                                   // the unit's gate-0 lane; + 8*g = gate g
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + (r & 3) + 8 * (r >> 2);
    const float pre = acc[0][0][r] + xv[r];
    const float sg = fast_sigmoid(gate == 2 ? 2.0f * pre : pre);
    const float av = gate == 2 ? 2.0f * sg - 1.0f : sg;
    const float gi = __shfl(av, src, 64), gf = __shfl(av, src + 8, 64);
    const float gg = __shfl(av, src + 16, 64), go = __shfl(av, src + 24, 64);
    float cn = cp[r], hn = hp[r];
    if (act[r]) {
      cn = gf * cp[r] + gi * gg;
      hn = go * fast_tanh(cn);
    }
    if (row < B && uok) {
      gates[(int64_t)row * 4 * H + gate * H + unit] = act[r] ? av : 0.f;
      if (gate == 0) {
        c_out[(int64_t)row * H + unit] = cn;
        h_out[(int64_t)row * H + unit] = hn;
        if (c_final) c_final[(int64_t)row * a.cf_ld + unit] = cn;
      }
    }
  }
}

// ------------------------------------------------------------------ backward step
// Staging shape: the four gate slices of the A operand (dgates rows, k inside gate s) are stacked as a 256-row
// type R image, the four slices of the B operand (W_hh rows s*H + k, 32 columns) side by side as a 128-column type C
// image -- exactly the operand images of a 256 x 128 tile, so the engine's loader loop, ring and barriers are reused
// as they are; only the MFMA side differs (each wave multiplies ITS slice pair, not the whole 256 x 128 product).
using CfgB = TileCfg<256, 128, 4, 2, 4, 2>;
constexpr int LB_BM = 64, LB_BN = 32;

template <int NV, int LT>
struct LstmBwdA {   // type R: image row = 64 * s + r  ->  dgates[m0 + r][s*H + k]
  struct Params { const float* dg; int B, H; };
  struct Raw { float4 v[NV]; };
  static constexpr bool kTypeR = true;
  const float* base;
  uint32_t voff[NV];
  int H;
  __device__ __forceinline__ void init(const Params& q, int m0, int tid, int /*ks0*/) {
    H = q.H;
    base = q.dg + (int64_t)m0 * 4 * q.H;
    const int c4 = 4 * StageMap<LT>::r_chunk(tid);
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int ir = StageMap<LT>::r_row(tid, p);      // 0..255
      const int s = ir >> 6, r = ir & 63;
      voff[p] = m0 + r < q.B ? (uint32_t)(r * 4 * q.H + s * q.H + c4) * 4u : BUF_OOB;
    }
  }
  __device__ __forceinline__ void issue(int ks, Raw& r) {
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(base + ks * BK, ks * BK < H ? BUF_OOB : 0u);
#pragma unroll
    for (int p = 0; p < NV; ++p) r.v[p] = buf_load16(rs, voff[p]);
  }
  __device__ __forceinline__ void finish(const Raw& r, float4 (&o)[NV]) const {
#pragma unroll
    for (int p = 0; p < NV; ++p) o[p] = r.v[p];
  }
};

template <int NV, int LT>
struct LstmBwdB {   // type C: image column = 32 * s + c  ->  W_hh[s*H + k][n0 + c]
  struct Params { const float* w; int H; };
  struct Raw { float4 v[NV]; };
  static constexpr bool kTypeR = false;
  const float* base;
  uint32_t voff[NV];
  int H;
  __device__ __forceinline__ void init(const Params& q, int n0, int tid, int /*ks0*/) {
    H = q.H;
    base = q.w + n0;
    const int kr = StageMap<LT>::c_krow(tid);
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int ch = StageMap<LT>::c_chunk(tid, p);    // 0..31: slice ch >> 3, columns 4 * (ch & 7) ..
      const int s = ch >> 3, c = 4 * (ch & 7);
      voff[p] = (uint32_t)((s * q.H + kr) * q.H + c) * 4u;     // n0 + c < H (H % 32 == 0)
    }
  }
  __device__ __forceinline__ void issue(int ks, Raw& r) {
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(base + (int64_t)ks * BK * H, ks * BK < H ? BUF_OOB : 0u);
#pragma unroll
    for (int p = 0; p < NV; ++p) r.v[p] = buf_load16(rs, voff[p]);
  }
  __device__ __forceinline__ void finish(const Raw& r, float4 (&o)[NV]) const {
#pragma unroll
    for (int p = 0; p < NV; ++p) o[p] = r.v[p];
  }
};

// The cell backward of one (sample, unit) at time t given dh = d loss / d h_t and dc = d loss / d c_t (both in
// place): writes the four pre-activation gradients, turns dc into d loss / d c_{t-1}.  vqa_lstm_cell_bwd's arithmetic.
__device__ __forceinline__ void cell_bwd_elem(bool active, float gi, float gf, float gg, float go, float c_in,
                                              float c_out, float dhv, float& dcv, float (&dgo)[4]) {
  dgo[0] = dgo[1] = dgo[2] = dgo[3] = 0.f;
  if (active) {
    const float tc = fast_tanh(c_out);
    const float dct = dcv + dhv * go * (1.f - tc * tc);
    dgo[0] = dct * gg * gi * (1.f - gi);
    dgo[1] = dct * c_in * gf * (1.f - gf);
    dgo[2] = dct * gi * (1.f - gg * gg);
    dgo[3] = dhv * tc * go * (1.f - go);
    dcv = dct * gf;
  }
}

// grid = ndir * tiles_m * tiles_n (tiles of 64 x 32), one workgroup per CU (105 KB LDS): 8 MFMA waves + 4 loader waves.
// Step n (0 .. T-2): t = bwd_time(n) is the time whose dgates are multiplied, t2 = bwd_time(n+1) the time whose cell
// is differentiated in the epilogue.
__global__ __launch_bounds__(CfgB::THREADS, CfgB::MIN_WAVES) void lstm_step_bwd_kernel(SeqArgs a, int n, int tiles_m,
                                                                                    int tiles_n) {
  using Cfg = CfgB;
  using AL = LstmBwdA<Cfg::NVA, Cfg::LT>;
  using BL = LstmBwdB<Cfg::NVB, Cfg::LT>;
  using SL = SmemLayout<Cfg, true, false>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = a.B, T = a.T, H = a.H;
  const int tiles = tiles_m * tiles_n;
  const int lid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int dir = lid / tiles, u = lid - dir * tiles;
  const int nt = u / tiles_m, mt = u - nt * tiles_m;       // row tile fastest: the row tiles of a W_hh slice are neighbours
  const SeqDir D = dir ? a.d[1] : a.d[0];
  const int t = bwd_time(D.reverse, n, T), t2 = bwd_time(D.reverse, n + 1, T);
  const int m0 = mt * LB_BM, n0 = nt * LB_BN;
  const int nk = H / BK;
  const float* dg_t = D.dgates + (int64_t)t * B * 4 * H;

  if (is_loader_wave<Cfg>()) {
    AL al; BL bl;
    al.init(typename AL::Params{dg_t, B, H}, m0, loader_tid<Cfg>(), 0);
    bl.init(typename BL::Params{D.w_hh, H}, n0, loader_tid<Cfg>(), 0);
    loader_loop<Cfg>(al, bl, 0, nk, smem);
  } else {
    // MFMA role: wave = (row half wm, gate slice wk); same barrier protocol as mfma_loop_eb (one per K-step, in
    // front of the last two fragment groups, so the stage returns to the loaders half a K-step early)
    const int wm = wave & 1, wk = wave >> 1;
    const int l31 = lane & 31, h = lane >> 5;
    constexpr int CSB = LdsImage<Cfg::BN>::CS;
    const float* const As0 = smem + ((64 * wk + 32 * wm + l31) * LDS_RS + 4 * h);
    const float* const Bs0 = smem + 2 * SL::ABUF + (4 * h * CSB + 32 * wk + l31);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float fa[4][4], fb[4][4];
    auto fetch = [&](const float* ap, const float* bp, int g, int buf) {
      const float4 v = *reinterpret_cast<const float4*>(ap + 8 * g);
      fa[buf][0] = v.x; fa[buf][1] = v.y; fa[buf][2] = v.z; fa[buf][3] = v.w;
#pragma unroll
      for (int q = 0; q < 4; ++q) fb[buf][q] = bp[(8 * g + q) * CSB];
    };
    auto mma = [&](int buf) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][q], fb[buf][q], acc, 0, 0, 0);
    };
    __builtin_amdgcn_s_setprio(VQA_PRIO_MFMA);
    __syncthreads();
    fetch(As0, Bs0, 0, 0);
    fetch(As0, Bs0, 1, 1);
    for (int ks = 0; ks < nk; ++ks) {
      const int cur = ks & 1;
      const float* const Ac = As0 + cur * SL::ABUF;
      const float* const Bc = Bs0 + cur * SL::BBUF;
      const float* const An = As0 + (cur ^ 1) * SL::ABUF;
      const float* const Bn = Bs0 + (cur ^ 1) * SL::BBUF;
      fetch(Ac, Bc, 2, 2);
      __builtin_amdgcn_sched_barrier(0);
      mma(0);
      __builtin_amdgcn_sched_barrier(0);
      fetch(Ac, Bc, 3, 3);
      __builtin_amdgcn_sched_barrier(0);
      mma(1);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      if (ks + 1 < nk) fetch(An, Bn, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma(2);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < nk) fetch(An, Bn, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(3);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    // partial tile of this wave -> comb[wk][row][33], re-using the staging memory: every fragment read of every
    // MFMA wave and every stage write of the loaders precede the last K-step barrier, so it is quiescent here
    float* comb = smem;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      comb[(wk * 64 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * h) * 33 + l31] = acc[r];
  }
  __syncthreads();          // all 12 waves
  if (tid >= Cfg::MFMA_THREADS) return;
  // ---- epilogue: dh_prev = sum of the four gate-slice partials (+ pass-through), then the cell backward at t2
  const float* comb = smem;
  const int col = tid & 31, r0 = tid >> 5;            // 16 row groups x 32 columns; rows r0 + 16 i
  const int j = n0 + col;
  const int64_t BH = (int64_t)B * H;
  const float* g2 = D.gates + (int64_t)t2 * B * 4 * H;
  float* dg2 = D.dgates + (int64_t)t2 * B * 4 * H;
  const float* c_in2 = D.Cs + slot_in(D.reverse, t2) * BH;
  const float* c_out2 = D.Cs + slot_out(D.reverse, t2) * BH;
  float gsum[4], dhv[4], dcv[4], gi[4], gf[4], gg[4], go[4], ci[4], co[4];
  int64_t ql[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {          // every load first: one batch in flight, stores afterwards
    const int row = r0 + 16 * i;
    const int b = m0 + row < B ? m0 + row : 0;
    gsum[i] = (comb[(0 * 64 + row) * 33 + col] + comb[(1 * 64 + row) * 33 + col]) +
              (comb[(2 * 64 + row) * 33 + col] + comb[(3 * 64 + row) * 33 + col]);
    const int64_t e = (int64_t)b * H + j, e4 = (int64_t)b * 4 * H + j;
    dhv[i] = D.dh[e]; dcv[i] = D.dc[e];
    gi[i] = g2[e4]; gf[i] = g2[e4 + H]; gg[i] = g2[e4 + 2 * H]; go[i] = g2[e4 + 3 * H];
    ci[i] = c_in2[e]; co[i] = c_out2[e];
    ql[i] = a.q_len[b];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = r0 + 16 * i;
    if (m0 + row >= B) continue;
    const int b = m0 + row;
    const int64_t e = (int64_t)b * H + j, e4 = (int64_t)b * 4 * H + j;
    // finished at t: h_t = h_{t-1}, so the incoming gradient passes through; its dgates row was zero
    const float d = gsum[i] + ((int64_t)t < ql[i] ? 0.f : dhv[i]);
    float dgo[4];
    float dcn = dcv[i];
    cell_bwd_elem((int64_t)t2 < ql[i], gi[i], gf[i], gg[i], go[i], ci[i], co[i], d, dcn, dgo);
    dg2[e4] = dgo[0]; dg2[e4 + H] = dgo[1]; dg2[e4 + 2 * H] = dgo[2]; dg2[e4 + 3 * H] = dgo[3];
    D.dc[e] = dcn;
    D.dh[e] = d;
  }
}

// First BPTT step: the cell backward at t = bwd_time(0) with dh = the caller's dh (zeros: only c_n is used by the
// model) and dc = gradient w.r.t. the final cell state.  grid (blocks, ndir).
__global__ void lstm_bwd_first_kernel(SeqArgs a) {
  const SeqDir D = blockIdx.y ? a.d[1] : a.d[0];
  const int B = a.B, T = a.T, H = a.H;
  const int t = bwd_time(D.reverse, 0, T);
  const int64_t BH = (int64_t)B * H;
  const float* g = D.gates + (int64_t)t * B * 4 * H;
  float* dg = D.dgates + (int64_t)t * B * 4 * H;
  const float* c_in = D.Cs + slot_in(D.reverse, t) * BH;
  const float* c_out = D.Cs + slot_out(D.reverse, t) * BH;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < BH; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / H), j = (int)(i - (int64_t)b * H);
    const int64_t e4 = (int64_t)b * 4 * H + j;
    float dgo[4];
    float dcn = D.dc[i];
    cell_bwd_elem((int64_t)t < a.q_len[b], g[e4], g[e4 + H], g[e4 + 2 * H], g[e4 + 3 * H], c_in[i], c_out[i], D.dh[i],
                  dcn, dgo);
    dg[e4] = dgo[0]; dg[e4 + H] = dgo[1]; dg[e4 + 2 * H] = dgo[2]; dg[e4 + 3 * H] = dgo[3];
    D.dc[i] = dcn;
  }
}

// ------------------------------------------------------------------ host side: launch plans and the graph cache
struct SeqPlan {
  const void* step_kernel;
  dim3 grid, block;
  size_t lds;
  int tiles_m, tiles_n;
  int first, last;          // step indices [first, last)
};

static SeqPlan plan_fwd(const SeqArgs& a) {
  using SL = SmemLayout<CfgL, true, true>;
  SeqPlan p;
  p.step_kernel = reinterpret_cast<const void*>(lstm_step_fwd_kernel);
  p.tiles_m = (a.B + CfgL::BM - 1) / CfgL::BM;
  p.tiles_n = (4 * a.H + CfgL::BN - 1) / CfgL::BN;
  p.grid = dim3(a.ndir * p.tiles_m * p.tiles_n);
  p.block = dim3(CfgL::THREADS);
  p.lds = SL::BYTES;
  p.first = 0; p.last = a.T;
  return p;
}
static SeqPlan plan_bwd(const SeqArgs& a) {
  using SL = SmemLayout<CfgB, true, false>;
  SeqPlan p;
  p.step_kernel = reinterpret_cast<const void*>(lstm_step_bwd_kernel);
  p.tiles_m = (a.B + LB_BM - 1) / LB_BM;
  p.tiles_n = a.H / LB_BN;
  p.grid = dim3(a.ndir * p.tiles_m * p.tiles_n);
  p.block = dim3(CfgB::THREADS);
  p.lds = SL::BYTES;
  p.first = 0; p.last = a.T - 1;
  return p;
}
static dim3 first_grid(const SeqArgs& a) {
  int64_t blocks = ((int64_t)a.B * a.H + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  return dim3((unsigned)blocks, (unsigned)a.ndir);
}

struct GraphKey {
  int kind, dev;
  SeqArgs a;
  bool operator<(const GraphKey& o) const { return memcmp(this, &o, sizeof(GraphKey)) < 0; }
};
struct GraphEntry { hipGraph_t graph; hipGraphExec_t exec; uint64_t stamp; };
static std::mutex g_graph_mu;
static std::map<GraphKey, GraphEntry> g_graphs;
static uint64_t g_graph_clock = 0;
static int g_graph_hits = 0, g_graph_builds = 0, g_graph_eager = 0;
// (kind, ndir, B, T, H) -> consecutive calls of that shape that found no cached graph (their buffers had moved)
static std::map<std::array<int, 5>, int> g_graph_miss_streak;
constexpr size_t kGraphCap = 32;
constexpr int kMissStreakLimit = 8;

static int build_graph(int kind, const SeqArgs& a, const SeqPlan& p, GraphEntry* out) {
  hipGraph_t g;
  int rc = check_hip(hipGraphCreate(&g, 0), "hipGraphCreate");
  if (rc) return rc;
  hipGraphNode_t prev = nullptr;
  SeqArgs args = a;
  auto add = [&](const void* func, dim3 grid, dim3 block, size_t lds, void** kargs) -> int {
    hipKernelNodeParams np;
    memset(&np, 0, sizeof(np));
    np.func = const_cast<void*>(func);
    np.gridDim = grid; np.blockDim = block; np.sharedMemBytes = (unsigned)lds;
    np.kernelParams = kargs; np.extra = nullptr;
    hipGraphNode_t node;
    int r = check_hip(hipGraphAddKernelNode(&node, g, prev ? &prev : nullptr, prev ? 1 : 0, &np), "hipGraphAddKernelNode");
    if (r) return r;
    prev = node;
    return VQA_OK;
  };
  if (kind == 1) {
    void* kargs[] = {&args};
    rc = add(reinterpret_cast<const void*>(lstm_bwd_first_kernel), first_grid(a), dim3(256), 0, kargs);
    if (rc) { hipGraphDestroy(g); return rc; }
  }
  for (int n = p.first; n < p.last; ++n) {
    int nn = n, tm = p.tiles_m, tn = p.tiles_n;
    void* kargs[] = {&args, &nn, &tm, &tn};
    rc = add(p.step_kernel, p.grid, p.block, p.lds, kargs);
    if (rc) { hipGraphDestroy(g); return rc; }
  }
  hipGraphExec_t exec;
  rc = check_hip(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0), "hipGraphInstantiate");
  if (rc) { hipGraphDestroy(g); return rc; }
  out->graph = g; out->exec = exec; out->stamp = 0;
  return VQA_OK;
}

static int run_sequence(int kind, const SeqArgs& a, int use_graph, hipStream_t s) {
  const SeqPlan p = kind == 0 ? plan_fwd(a) : plan_bwd(a);
  int rc = ensure_dyn_smem(p.step_kernel, (int)p.lds, kind == 0 ? "attr(lstm_step_fwd)" : "attr(lstm_step_bwd)");
  if (rc) return rc;
  if (use_graph) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) use_graph = 0;   // the caller captures itself
  }
  if (use_graph) {
    GraphKey key;
    memset(&key, 0, sizeof(key));
    key.kind = kind;
    hipGetDevice(&key.dev);
    key.a = a;
    std::lock_guard<std::mutex> lk(g_graph_mu);
    auto it = g_graphs.find(key);
    int& streak = g_graph_miss_streak[std::array<int, 5>{kind, a.ndir, a.B, a.T, a.H}];
    if (it == g_graphs.end()) {
      // a caller whose buffers move on every call would build a graph per call: after 8 consecutive misses of one
      // shape its calls are served by plain launches; every 64th of them tries a graph again (a loop that settles
      // later returns to replays)
      ++streak;
      if (streak > kMissStreakLimit && (streak & 63) != 0) {
        use_graph = 0;
        ++g_graph_eager;
      } else {
        GraphEntry e;
        rc = build_graph(kind, a, p, &e);
        if (rc) return rc;
        if (g_graphs.size() >= kGraphCap) {     // drop the least recently used graph
          auto old = g_graphs.begin();
          for (auto j = g_graphs.begin(); j != g_graphs.end(); ++j) if (j->second.stamp < old->second.stamp) old = j;
          hipGraphExecDestroy(old->second.exec);
          hipGraphDestroy(old->second.graph);
          g_graphs.erase(old);
        }
        it = g_graphs.emplace(key, e).first;
        ++g_graph_builds;
      }
    } else {
      streak = 0;
      ++g_graph_hits;
    }
    if (use_graph) {
      it->second.stamp = ++g_graph_clock;
      return check_hip(hipGraphLaunch(it->second.exec, s), "hipGraphLaunch(lstm sequence)");
    }
  }
  if (kind == 1) {
    hipLaunchKernelGGL(lstm_bwd_first_kernel, first_grid(a), dim3(256), 0, s, a);
    rc = check_hip(hipGetLastError(), "lstm_bwd_first launch");
    if (rc) return rc;
  }
  for (int n = p.first; n < p.last; ++n) {
    if (kind == 0) hipLaunchKernelGGL(lstm_step_fwd_kernel, p.grid, p.block, p.lds, s, a, n, p.tiles_m, p.tiles_n);
    else hipLaunchKernelGGL(lstm_step_bwd_kernel, p.grid, p.block, p.lds, s, a, n, p.tiles_m, p.tiles_n);
    rc = check_hip(hipGetLastError(), kind == 0 ? "lstm_step_fwd launch" : "lstm_step_bwd launch");
    if (rc) return rc;
  }
  return VQA_OK;
}

static int make_args(const char* fn, const vqa_lstm_dir_t* dirs, int ndir, const int64_t* q_len, int B, int T, int H,
                     int64_t cf_ld, bool backward, SeqArgs* out) {
  VQA_REQUIRE(dirs && q_len && (ndir == 1 || ndir == 2), "%s: dirs / q_len null or ndir=%d not 1 or 2", fn, ndir);
  VQA_REQUIRE(B > 0 && T > 0 && vqa_lstm_step_supported(H), "%s: B=%d T=%d must be positive, H=%d a positive multiple of %d",
              fn, B, T, H, BK);
  VQA_REQUIRE((int64_t)4 * H * H * 4 < 0xffff0000LL && (int64_t)B * 4 * H * 4 < 0xffff0000LL,
              "%s: W_hh or one step's gate tensor reaches 4 GiB", fn);
  memset(out, 0, sizeof(SeqArgs));
  for (int d = 0; d < ndir; ++d) {
    const vqa_lstm_dir_t& s = dirs[d];
    VQA_REQUIRE(s.w_hh && s.gates && s.Hs && s.Cs, "%s: direction %d has a null w_hh / gates / Hs / Cs", fn, d);
    VQA_REQUIRE(((uintptr_t)s.w_hh % 16) == 0 && ((uintptr_t)s.Hs % 16) == 0, "%s: w_hh / Hs must be 16-byte aligned", fn);
    if (backward) {
      VQA_REQUIRE(s.dgates && s.dh && s.dc, "%s: direction %d has a null dgates / dh / dc", fn, d);
      VQA_REQUIRE(((uintptr_t)s.dgates % 16) == 0, "%s: dgates must be 16-byte aligned", fn);
    } else {
      VQA_REQUIRE(s.xg, "%s: direction %d has a null xg", fn, d);
    }
    SeqDir& o = out->d[d];
    o.w_hh = s.w_hh; o.xg = s.xg; o.gates = s.gates; o.Hs = s.Hs; o.Cs = s.Cs; o.c_final = s.c_final;
    o.dgates = s.dgates; o.dh = s.dh; o.dc = s.dc; o.reverse = s.reverse ? 1 : 0;
  }
  out->q_len = q_len; out->cf_ld = cf_ld; out->ndir = ndir; out->B = B; out->T = T; out->H = H;
  return VQA_OK;
}

}  // namespace vqa

using namespace vqa;

extern "C" {

int vqa_lstm_step_supported(int H) { return (H > 0 && H % BK == 0) ? 1 : 0; }

int vqa_lstm_seq_fwd(const vqa_lstm_dir_t* dirs, int ndir, const int64_t* q_len, int B, int T, int H, int64_t cf_ld,
                     int use_graph, vqa_stream_t stream) {
  SeqArgs a;
  int rc = make_args("vqa_lstm_seq_fwd", dirs, ndir, q_len, B, T, H, cf_ld, false, &a);
  if (rc) return rc;
  set_launch_tag(0);
  ProfScope prof(VQA_K_LSTM_SEQ, (hipStream_t)stream);
  return run_sequence(0, a, use_graph, (hipStream_t)stream);
}

int vqa_lstm_seq_bwd(const vqa_lstm_dir_t* dirs, int ndir, const int64_t* q_len, int B, int T, int H, int use_graph,
                     vqa_stream_t stream) {
  SeqArgs a;
  int rc = make_args("vqa_lstm_seq_bwd", dirs, ndir, q_len, B, T, H, 0, true, &a);
  if (rc) return rc;
  set_launch_tag(1);
  ProfScope prof(VQA_K_LSTM_SEQ, (hipStream_t)stream);
  return run_sequence(1, a, use_graph, (hipStream_t)stream);
}

int vqa_lstm_graph_stats(int* replays, int* builds, int* plain) {
  std::lock_guard<std::mutex> lk(g_graph_mu);
  if (replays) *replays = g_graph_hits;
  if (builds) *builds = g_graph_builds;
  if (plain) *plain = g_graph_eager;
  return (int)g_graphs.size();
}

}  // extern "C"
