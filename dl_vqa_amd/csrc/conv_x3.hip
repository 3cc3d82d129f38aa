// Convolution blocks in fp32 on the bf16 matrix cores (x3_core.hpp: exact three-way bf16 split, six partial products,
// fp32 accumulate).  Same tensors, layouts, loaders and epilogues as conv.hip -- only the K loop differs -- so the
// entry points take exactly the arguments of their conv.hip counterparts.  Reference: models/model.py:72-84.
#include "x3_core.hpp"

namespace vqa {

int colsum_launch(const float* x, int64_t ld, const uint8_t* mask, int64_t rows, int cols, float* out,
                  int accumulate, float* ws, int64_t ws_bytes, hipStream_t s);
int64_t colsum_ws_bytes(int64_t rows, int cols);

#include "conv_device.inc"
#include "conv_host.inc"

#ifndef VQA_X3_PF
#define VQA_X3_PF 2
#endif
// one workgroup per CU (240 bytes of LDS per tile row and stage): 4 MFMA waves + 4 loader waves, 256 VGPRs each
using CfgX = TileCfg<192, 128, 2, 2, 4, VQA_X3_PF>;      // MFMA waves of 96 x 64
using CfgXn = TileCfg<256, 64, 4, 1, 4, VQA_X3_PF>;      // 64 output columns (conv1 dgrad): MFMA waves of 64 x 64

template <class Cfg>
__global__ __launch_bounds__(Cfg::THREADS, 2) void conv_fwd_x3_kernel(typename ConvFwdA<Cfg::NVA, Cfg::LT, true>::Params pa,
                                                                      typename PlainCx<Cfg::NVB, Cfg::LT>::Params pb,
                                                                      const float* __restrict__ bias, float* pooled,
                                                                      uint8_t* amax, int Co, int tiles_m, int tiles_n, int nk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const TileCoord tc = tile_coord(tiles_m, tiles_n);
  const int m0 = tc.mt * Cfg::BM, n0 = tc.nt * Cfg::BN;
  using AL = ConvFwdA<Cfg::NVA, Cfg::LT, true>;
  using BL = PlainCx<Cfg::NVB, Cfg::LT>;
  f32x16 acc[Cfg::TM][Cfg::TN];
  acc_zero<Cfg>(acc);
  if (!gemm_mainloop_x<Cfg, AL, BL>(
          [&](AL& al, BL& bl) {
            al.init(pa, m0, loader_tid<Cfg>(), 0);
            bl.init(pb, n0, loader_tid<Cfg>(), 0);
          },
          acc, 0, nk, smem))
    return;
  conv_pool_epilogue<Cfg>(acc, bias, pooled, amax, pa.nWin, Co, m0, n0, wm, wn, lane);
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::THREADS, 2) void conv_dgrad_x3_kernel(typename ConvDgradA<Cfg::NVA, Cfg::LT, true>::Params pa,
                                                                        typename PlainCx<Cfg::NVB, Cfg::LT>::Params pb,
                                                                        float* dx, int CiP, int tiles_m, int tiles_n, int nk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const TileCoord tc = tile_coord(tiles_m, tiles_n);
  const int m0 = tc.mt * Cfg::BM, n0 = tc.nt * Cfg::BN;
  using AL = ConvDgradA<Cfg::NVA, Cfg::LT, true>;
  using BL = PlainCx<Cfg::NVB, Cfg::LT>;
  f32x16 acc[Cfg::TM][Cfg::TN];
  acc_zero<Cfg>(acc);
  if (!gemm_mainloop_x<Cfg, AL, BL>(
          [&](AL& al, BL& bl) {
            al.init(pa, m0, loader_tid<Cfg>(), 0);
            bl.init(pb, n0, loader_tid<Cfg>(), 0);
          },
          acc, 0, nk, smem))
    return;
  store_acc_tiles<Cfg>(acc, dx, CiP, pa.rows, CiP, m0, n0, wm, wn, lane);
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::THREADS, 2) void conv_wgrad_x3_kernel(typename WgradA<Cfg::NVA, Cfg::LT, true>::Params pa,
                                                                        typename WgradB<Cfg::NVB, Cfg::LT, true>::Params pb,
                                                                        float* slab, int tiles_m, int tiles_n, int nk,
                                                                        int ks_per_split) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const TileCoord tc = tile_coord(tiles_m, tiles_n);
  const int m0 = tc.mt * Cfg::BM, n0 = tc.nt * Cfg::BN;
  const int ks0 = tc.split * ks_per_split;
  const int ks1 = min(nk, ks0 + ks_per_split);
  using AL = WgradA<Cfg::NVA, Cfg::LT, true>;
  using BL = WgradB<Cfg::NVB, Cfg::LT, true>;
  f32x16 acc[Cfg::TM][Cfg::TN];
  acc_zero<Cfg>(acc);
  if (!gemm_mainloop_x<Cfg, AL, BL>(
          [&](AL& al, BL& bl) {
            al.init(pa, m0, loader_tid<Cfg>(), ks0);
            bl.init(pb, n0, loader_tid<Cfg>(), ks0);
          },
          acc, ks0, ks1, smem))
    return;
  const int Co = pa.g.Co;
  store_acc_tiles<Cfg>(acc, slab + (int64_t)tc.split * pa.KI * Co, Co, pa.KI, Co, m0, n0, wm, wn, lane);
}

template <class Cfg>
static int launch_fwd_x3(const float* x, const void* wf, const float* bias, float* pooled, uint8_t* amax,
                         const ConvGeom& g, hipStream_t s) {
  using SL = SmemLayoutX<Cfg, true, false>;
  const int nWin = g.B * g.Hp * g.Wp, K = 9 * g.CiP;
  typename ConvFwdA<Cfg::NVA, Cfg::LT, true>::Params pa{x, g.H, g.W, g.CiP, g.Hp, g.Wp, g.stride, nWin, K};
  typename PlainCx<Cfg::NVB, Cfg::LT>::Params pb{wf, g.Co, g.Co, K, (int64_t)K * g.Co};
  const int tiles_m = (4 * nWin + Cfg::BM - 1) / Cfg::BM, tiles_n = (g.Co + Cfg::BN - 1) / Cfg::BN;
  auto kern = conv_fwd_x3_kernel<Cfg>;
  { int rc = set_smem(kern, SL::BYTES, "attr(conv_fwd_x3)"); if (rc) return rc; }
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(Cfg::THREADS), SL::BYTES, s, pa, pb, bias, pooled, amax, g.Co,
                     tiles_m, tiles_n, K / BK);
  return check_hip(hipGetLastError(), "conv_fwd_x3 launch");
}

template <class Cfg>
static int launch_dgrad_x3(const float* dp, const uint8_t* am, const void* wd, float* dx, const ConvGeom& g,
                           hipStream_t s) {
  using SL = SmemLayoutX<Cfg, true, false>;
  const int rows = g.B * g.H * g.W, K = 9 * g.Co;
  typename ConvDgradA<Cfg::NVA, Cfg::LT, true>::Params pa{dp, am, g.H, g.W, g.Hp, g.Wp, g.Co, g.stride, rows, K};
  typename PlainCx<Cfg::NVB, Cfg::LT>::Params pb{wd, g.CiP, g.CiP, K, (int64_t)K * g.CiP};
  const int tiles_m = (rows + Cfg::BM - 1) / Cfg::BM, tiles_n = (g.CiP + Cfg::BN - 1) / Cfg::BN;
  auto kern = conv_dgrad_x3_kernel<Cfg>;
  { int rc = set_smem(kern, SL::BYTES, "attr(conv_dgrad_x3)"); if (rc) return rc; }
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(Cfg::THREADS), SL::BYTES, s, pa, pb, dx, g.CiP, tiles_m,
                     tiles_n, K / BK);
  return check_hip(hipGetLastError(), "conv_dgrad_x3 launch");
}

struct WgradPlanX { int tiles_m, tiles_n, nk, splits, ks_per_split, Mtot, KI; };
static WgradPlanX plan_wgrad_x3(const ConvGeom& g) {
  WgradPlanX p;
  p.KI = 9 * g.CiP;
  p.Mtot = g.B * 2 * g.Hp * 2 * g.Wp;
  p.tiles_m = (p.KI + CfgX::BM - 1) / CfgX::BM;
  p.tiles_n = (g.Co + CfgX::BN - 1) / CfgX::BN;
  p.nk = (p.Mtot + BK - 1) / BK;
  int splits = 256 / (p.tiles_m * p.tiles_n);       // one workgroup per CU: at most 256 resident
  if (splits < 1) splits = 1;
  const int max_splits = p.nk / 8 > 1 ? p.nk / 8 : 1;
  if (splits > max_splits) splits = max_splits;
  p.ks_per_split = (p.nk + splits - 1) / splits;
  p.splits = (p.nk + p.ks_per_split - 1) / p.ks_per_split;
  return p;
}

// the operand split on its own (tests): planes of n bf16 each
__global__ void x3_split_kernel(const float4* x, uint2* hi, uint2* mid, uint2* lo, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) split4(x[i], hi[i], mid[i], lo[i], split_consts());
}

static bool x3_conv_ok(int CiP, int Co, int Wp) { return CiP % BK == 0 && Co % BK == 0 && 2 * Wp >= BK; }

}  // namespace vqa

using namespace vqa;

extern "C" {

int vqa_x3_split(const float* x, void* hi, void* mid, void* lo, int64_t n, vqa_stream_t stream) {
  VQA_REQUIRE(x && hi && mid && lo && n > 0 && n % 4 == 0, "vqa_x3_split: bad args (n=%lld must be a multiple of 4)", (long long)n);
  hipLaunchKernelGGL(x3_split_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(x), static_cast<uint2*>(hi), static_cast<uint2*>(mid),
                     static_cast<uint2*>(lo), n / 4);
  return check_hip(hipGetLastError(), "x3_split launch");
}

int vqa_conv3x3_x3_supported(int H, int W, int CiP, int Co, int stride) {
  const ConvGeom g = make_geom(1, H, W, CiP, Co, stride);
  return (g.Hp > 0 && g.Wp > 0 && x3_conv_ok(CiP, Co, g.Wp)) ? 1 : 0;
}

int vqa_conv3x3_relu_pool_fwd_x3(const float* x, const void* wf, const float* bias, float* pooled, uint8_t* argmax,
                                 int B, int H, int W, int CiP, int Co, int stride, int tag, vqa_stream_t stream) {
  VQA_REQUIRE(x && wf && bias && pooled && argmax && B > 0, "vqa_conv3x3_relu_pool_fwd_x3: null pointer");
  VQA_REQUIRE(CiP % BK == 0, "vqa_conv3x3_relu_pool_fwd_x3: CiP=%d must be a multiple of %d", CiP, BK);
  const int chunk = batch_chunk(B, H, W, CiP, Co, stride);
  VQA_REQUIRE(chunk > 0, "vqa_conv3x3_relu_pool_fwd_x3: one %dx%dx%d image reaches 4 GiB", H, W, CiP);
  const ConvGeom g1 = make_geom(1, H, W, CiP, Co, stride);
  set_launch_tag(tag);
  ProfScope prof(VQA_K_CONV_FWD, (hipStream_t)stream);
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = B - b0 < chunk ? B - b0 : chunk;
    const int64_t xo = (int64_t)b0 * H * W * CiP, po = (int64_t)b0 * g1.Hp * g1.Wp * Co;
    const ConvGeom g = make_geom(nb, H, W, CiP, Co, stride);
    int rc = check_geom("vqa_conv3x3_relu_pool_fwd_x3", g);
    if (rc) return rc;
    rc = Co > 64 ? launch_fwd_x3<CfgX>(x + xo, wf, bias, pooled + po, argmax + po, g, (hipStream_t)stream)
                 : launch_fwd_x3<CfgXn>(x + xo, wf, bias, pooled + po, argmax + po, g, (hipStream_t)stream);
    if (rc) return rc;
  }
  return VQA_OK;
}

int vqa_conv3x3_dgrad_x3(const float* dpooled, const uint8_t* argmax, const void* wd, float* dx, int B, int H, int W,
                         int CiP, int Co, int stride, int tag, vqa_stream_t stream) {
  VQA_REQUIRE(dpooled && argmax && wd && dx && B > 0, "vqa_conv3x3_dgrad_x3: null pointer");
  VQA_REQUIRE(Co % BK == 0, "vqa_conv3x3_dgrad_x3: Co=%d must be a multiple of %d", Co, BK);
  const int chunk = batch_chunk(B, H, W, CiP, Co, stride);
  VQA_REQUIRE(chunk > 0, "vqa_conv3x3_dgrad_x3: one %dx%dx%d image reaches 4 GiB", H, W, CiP);
  const ConvGeom g1 = make_geom(1, H, W, CiP, Co, stride);
  set_launch_tag(tag);
  ProfScope prof(VQA_K_CONV_DGRAD, (hipStream_t)stream);
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = B - b0 < chunk ? B - b0 : chunk;
    const int64_t xo = (int64_t)b0 * H * W * CiP, po = (int64_t)b0 * g1.Hp * g1.Wp * Co;
    const ConvGeom g = make_geom(nb, H, W, CiP, Co, stride);
    int rc = check_geom("vqa_conv3x3_dgrad_x3", g);
    if (rc) return rc;
    rc = CiP > 64 ? launch_dgrad_x3<CfgX>(dpooled + po, argmax + po, wd, dx + xo, g, (hipStream_t)stream)
                  : launch_dgrad_x3<CfgXn>(dpooled + po, argmax + po, wd, dx + xo, g, (hipStream_t)stream);
    if (rc) return rc;
  }
  return VQA_OK;
}

int64_t vqa_conv3x3_wgrad_x3_workspace_bytes(int B, int H, int W, int CiP, int Co, int stride) {
  const ConvGeom g1 = make_geom(1, H, W, CiP, Co, stride);
  if (g1.Hp <= 0 || g1.Wp <= 0 || B <= 0) return 0;
  const int chunk = batch_chunk(B, H, W, CiP, Co, stride);
  if (chunk <= 0) return 0;
  int64_t parts = 0;
  for (int b0 = 0; b0 < B; b0 += chunk)
    parts += plan_wgrad_x3(make_geom(B - b0 < chunk ? B - b0 : chunk, H, W, CiP, Co, stride)).splits;
  return parts * (int64_t)9 * CiP * Co * 4 + colsum_ws_bytes((int64_t)B * g1.Hp * g1.Wp, Co);
}

int vqa_conv3x3_wgrad_x3(const float* x, const float* dpooled, const uint8_t* argmax, float* dw, float* dbias, int B,
                         int H, int W, int CiP, int Ci, int Co, int stride, float* workspace, int64_t workspace_bytes,
                         int tag, vqa_stream_t stream) {
  VQA_REQUIRE(x && dpooled && argmax && dw && dbias && workspace && B > 0, "vqa_conv3x3_wgrad_x3: null pointer");
  VQA_REQUIRE(Ci >= 1 && Ci <= CiP, "vqa_conv3x3_wgrad_x3: Ci=%d CiP=%d", Ci, CiP);
  const ConvGeom g1 = make_geom(1, H, W, CiP, Co, stride);
  VQA_REQUIRE(g1.Hp > 0 && g1.Wp > 0 && x3_conv_ok(CiP, Co, g1.Wp),
              "vqa_conv3x3_wgrad_x3: needs CiP, Co multiples of %d and 2*Wp >= %d (CiP=%d Co=%d Wp=%d)", BK, BK, CiP, Co, g1.Wp);
  const int chunk = batch_chunk(B, H, W, CiP, Co, stride);
  VQA_REQUIRE(chunk > 0, "vqa_conv3x3_wgrad_x3: one %dx%dx%d image reaches 4 GiB", H, W, CiP);
  const int64_t need = vqa_conv3x3_wgrad_x3_workspace_bytes(B, H, W, CiP, Co, stride);
  if (workspace_bytes < need) {
    set_error("vqa_conv3x3_wgrad_x3: workspace %lld < %lld", (long long)workspace_bytes, (long long)need);
    return VQA_ERR_WORKSPACE;
  }
  const int KI = 9 * CiP;
  int parts = 0;
  for (int b0 = 0; b0 < B; b0 += chunk)
    parts += plan_wgrad_x3(make_geom(B - b0 < chunk ? B - b0 : chunk, H, W, CiP, Co, stride)).splits;
  float* const colsum_ws = workspace + (int64_t)parts * KI * Co;
  hipStream_t s = (hipStream_t)stream;
  set_launch_tag(tag);
  ProfScope prof(VQA_K_CONV_WGRAD, s);
  int done = 0, rc;
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = B - b0 < chunk ? B - b0 : chunk;
    const ConvGeom g = make_geom(nb, H, W, CiP, Co, stride);
    rc = check_geom("vqa_conv3x3_wgrad_x3", g);
    if (rc) return rc;
    const WgradPlanX p = plan_wgrad_x3(g);
    using SL = SmemLayoutX<CfgX, false, false>;
    WgradGeom wg{g.H, g.W, g.CiP, g.Hp, g.Wp, g.Co, g.stride, p.Mtot};
    typename WgradA<CfgX::NVA, CfgX::LT, true>::Params pa{x + (int64_t)b0 * H * W * CiP, wg, p.KI};
    const int64_t po = (int64_t)b0 * g1.Hp * g1.Wp * Co;
    typename WgradB<CfgX::NVB, CfgX::LT, true>::Params pb{dpooled + po, argmax + po, wg};
    auto kern = conv_wgrad_x3_kernel<CfgX>;
    rc = set_smem(kern, SL::BYTES, "attr(conv_wgrad_x3)");
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n * p.splits), dim3(CfgX::THREADS), SL::BYTES, s, pa, pb,
                       workspace + (int64_t)done * KI * Co, p.tiles_m, p.tiles_n, p.nk, p.ks_per_split);
    rc = check_hip(hipGetLastError(), "conv_wgrad_x3 launch");
    if (rc) return rc;
    done += p.splits;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((KI * Co + 63) / 64), dim3(256), 0, s, workspace, dw, parts, KI, CiP, Ci,
                     Co);
  rc = check_hip(hipGetLastError(), "wgrad_reduce launch");
  if (rc) return rc;
  // bias gradient = sum of the pooled gradient over the windows whose ReLU was alive (arg-max byte != 4)
  const int64_t rows = (int64_t)B * g1.Hp * g1.Wp;
  return colsum_launch(dpooled, Co, argmax, rows, Co, dbias, 0, colsum_ws, colsum_ws_bytes(rows, Co), s);
}

}  // extern "C"
