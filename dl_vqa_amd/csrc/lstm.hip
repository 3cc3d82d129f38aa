// Fused LSTM time step: gates = x-projection + h_{t-1} W_hh^T (MFMA), then the cell, in ONE kernel.
//
// Reference: nn.LSTM inside questionNet (models/model.py:145-149, 159-164): per step and direction
//   [i f g o] = x_t W_ih^T + b_ih + h_{t-1} W_hh^T + b_hh;  c' = s(f) c + s(i) tanh(g);  h' = s(o) tanh(c'),
// with the packed-sequence rule that sample b only advances while t < q_len[b].
//
// The x half is one big GEMM over all T (vqa_gemm, xg [T*B][4H]); the recurrent half is this kernel:
//   * GEMM engine of gemm_core.hpp, 64x64 tiles: A = h_{t-1} [B][H] staged through LDS by the loader waves
//     (type R), B = W_hh rows GATHERED so that a workgroup's 64 output columns are the four gates of 16 hidden
//     units (column c of MFMA wave wn: gate c/8, unit 16*tile + 8*wn + c%8) -- W_hh keeps PyTorch's
//     [i|f|g|o] row layout, the gather is only an offset computed once per tile;
//   * epilogue: a lane holds one gate of (row, unit) for 16 rows; it adds xg, applies its own activation, the
//     four lanes of a unit exchange their gates (ds_bpermute), and c', h', the saved gate activations and (on
//     the last step) the final cell state are written directly -- the [B][4H] pre-activation tensor of the
//     unfused path is never materialised, and the separate cell launch is gone.
// Loads of the epilogue are issued as one batch before the per-element code (one vmcnt for loads and stores).
#include "gemm_core.hpp"

namespace vqa {

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

__global__ __launch_bounds__(CfgL::THREADS, CfgL::MIN_WAVES) void lstm_step_fwd_kernel(
    typename PlainR<CfgL::NVA, CfgL::LT>::Params pa, typename LstmWhhR<CfgL::NVB, CfgL::LT>::Params pb,
    const float* __restrict__ xg, const float* __restrict__ c_in, const int64_t* __restrict__ q_len, int t,
    float* gates, float* c_out, float* h_out, float* c_final, int64_t cf_ld, int B, int H, int tiles_m, int tiles_n) {
  using Cfg = CfgL;
  using AL = PlainR<Cfg::NVA, Cfg::LT>;
  using BL = LstmWhhR<Cfg::NVB, Cfg::LT>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const TileCoord tc = tile_coord(tiles_m, tiles_n, 1, 1);     // weight-stationary order: an XCD keeps its W_hh slice
  const int m0 = tc.mt * Cfg::BM, n0 = tc.nt * Cfg::BN;
  f32x16 acc[Cfg::TM][Cfg::TN];
  acc_zero<Cfg>(acc);
  if (!gemm_mainloop<Cfg, AL, BL>(
          [&](AL& al, BL& bl) {
            al.init(pa, m0, loader_tid<Cfg>(), 0);
            bl.init(pb, n0, loader_tid<Cfg>(), 0);
          },
          [](AL&, BL&) {}, acc, 0, H / BK, H, smem))
    return;
  // ---- cell epilogue
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
    hp[r] = pa.p[(int64_t)rr * pa.ld + uu];
    act[r] = (int64_t)t < q_len[rr];
  }
  const int src = lane & ~24;                                   // the unit's gate-0 lane; + 8*g = gate g
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + (r & 3) + 8 * (r >> 2);
    const float pre = acc[0][0][r] + xv[r];
    const float a = gate == 2 ? tanhf(pre) : sigmoidf_(pre);
    const float gi = __shfl(a, src, 64), gf = __shfl(a, src + 8, 64);
    const float gg = __shfl(a, src + 16, 64), go = __shfl(a, src + 24, 64);
    float cn = cp[r], hn = hp[r];
    if (act[r]) {
      cn = gf * cp[r] + gi * gg;
      hn = go * tanhf(cn);
    }
    if (row < B && uok) {
      gates[(int64_t)row * 4 * H + gate * H + unit] = act[r] ? a : 0.f;
      if (gate == 0) {
        c_out[(int64_t)row * H + unit] = cn;
        h_out[(int64_t)row * H + unit] = hn;
        if (c_final) c_final[(int64_t)row * cf_ld + unit] = cn;
      }
    }
  }
}

}  // namespace vqa

using namespace vqa;

extern "C" {

int vqa_lstm_step_supported(int H) { return (H > 0 && H % BK == 0) ? 1 : 0; }

int vqa_lstm_step_fwd(const float* h_in, const float* w_hh, const float* xg_t, const float* c_in, const int64_t* q_len,
                      int t, float* gates, float* c_out, float* h_out, float* c_final, int64_t cf_ld, int B, int H,
                      vqa_stream_t stream) {
  VQA_REQUIRE(h_in && w_hh && xg_t && c_in && q_len && gates && c_out && h_out, "vqa_lstm_step_fwd: null pointer");
  VQA_REQUIRE(B > 0 && vqa_lstm_step_supported(H), "vqa_lstm_step_fwd: H=%d must be a positive multiple of %d", H, BK);
  VQA_REQUIRE(((uintptr_t)h_in % 16) == 0 && ((uintptr_t)w_hh % 16) == 0, "vqa_lstm_step_fwd: h_in / w_hh must be 16-byte aligned");
  VQA_REQUIRE((int64_t)4 * H * H * 4 < 0xffff0000LL, "vqa_lstm_step_fwd: W_hh reaches 4 GiB");
  using Cfg = CfgL;
  using SL = SmemLayout<Cfg, true, true>;
  typename PlainR<Cfg::NVA, Cfg::LT>::Params pa{h_in, (int64_t)H, B, H};
  typename LstmWhhR<Cfg::NVB, Cfg::LT>::Params pb{w_hh, H};
  const int tiles_m = (B + Cfg::BM - 1) / Cfg::BM, tiles_n = (4 * H + Cfg::BN - 1) / Cfg::BN;
  {
    int rc = ensure_dyn_smem(reinterpret_cast<const void*>(lstm_step_fwd_kernel), SL::BYTES, "attr(lstm_step_fwd)");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(lstm_step_fwd_kernel, dim3(tiles_m * tiles_n), dim3(Cfg::THREADS), SL::BYTES, (hipStream_t)stream,
                     pa, pb, xg_t, c_in, q_len, t, gates, c_out, h_out, c_final, cf_ld, B, H, tiles_m, tiles_n);
  return check_hip(hipGetLastError(), "lstm_step_fwd launch");
}

}  // extern "C"
