// bf16 convolution entry points (BASELINE configs[3]); kernels in conv_bf16.inc, shared device code in conv_device.inc.
#include "bf16_core.hpp"

namespace vqa {

#include "conv_device.inc"
#include "conv_host.inc"
#include "conv_bf16.inc"

using Cfg128 = TileCfg<128, 128, 2, 2>;
using Cfg128x64 = TileCfg<128, 64, 2, 2>;
// 256-row tiles (8 MFMA waves + 4 loader waves, one workgroup per CU): a third fewer staged bytes per MFMA than two
// 128x128 workgroups -- on bf16 MFMA the kernels are bound by exactly those bytes (VQA_BIG_TILES=1 selects them)
using Cfg256x128 = TileCfg<256, 128, 4, 2, 4, 2>;
using Cfg256x64 = TileCfg<256, 64, 4, 2, 4, 2>;
#ifndef VQA_BF16_FWD_PF
#define VQA_BF16_FWD_PF 2
#endif
using Cfg128x64L8 = TileCfg<128, 64, 2, 2, 8, 2>;   // 8 loader waves: the bf16 routing (2.5 VALU per element) is the long pole of dgrad
using Cfg128F = TileCfg<128, 128, 2, 2, 4, VQA_BF16_FWD_PF>;     // forward: plain copies, a third ring slot fits the registers

// ------------------------------------------------------------------ bf16 path launchers (conv_bf16.inc)
using CfgWb = TileCfg<128, 128, 2, 2, 4, 2>;      // bf16 wgrad: both operands through the transpose-read image

template <class Cfg, bool OB>
static int launch_fwd_bf16(const void* x, const void* wfT, const float* bias, void* pooled, uint8_t* amax,
                           const ConvGeom& g, hipStream_t s) {
  using SL = SmemLayoutB<Cfg, true, true>;
  const int nWin = g.B * g.Hp * g.Wp, Kdw = 9 * g.CiP / 2;
  typename ConvFwdA<Cfg::NVA, Cfg::LT, true>::Params pa{static_cast<const float*>(x), g.H, g.W, g.CiP / 2, g.Hp, g.Wp,
                                                         g.stride, nWin, Kdw};
  typename PlainR<Cfg::NVB, Cfg::LT>::Params pb{static_cast<const float*>(wfT), Kdw, g.Co, Kdw};
  const int tiles_m = (4 * nWin + Cfg::BM - 1) / Cfg::BM, tiles_n = (g.Co + Cfg::BN - 1) / Cfg::BN;
  auto kern = conv_fwd_bf16_kernel<Cfg, OB>;
  int rc = set_smem(kern, SL::BYTES, "attr(conv_fwd_bf16)");
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(Cfg::THREADS), SL::BYTES, s, pa, pb, bias, pooled, amax, g.Co,
                     tiles_m, tiles_n, Kdw / BK);
  return check_hip(hipGetLastError(), "conv_fwd_bf16 launch");
}

template <class Cfg, bool OB>
static int launch_dgrad_bf16(const void* dp, const uint8_t* am, const void* wdT, void* dx, const ConvGeom& g,
                             hipStream_t s) {
  using SL = SmemLayoutB<Cfg, true, true>;
  const int rows = g.B * g.H * g.W, Kdw = 9 * g.Co / 2;
  typename ConvDgradAb<Cfg::NVA, Cfg::LT>::Params pa{dp, am, g.H, g.W, g.Hp, g.Wp, g.Co, g.stride, rows};
  typename PlainR<Cfg::NVB, Cfg::LT>::Params pb{static_cast<const float*>(wdT), Kdw, g.CiP, Kdw};
  const int tiles_m = (rows + Cfg::BM - 1) / Cfg::BM, tiles_n = (g.CiP + Cfg::BN - 1) / Cfg::BN;
  auto kern = conv_dgrad_bf16_kernel<Cfg, OB>;
  int rc = set_smem(kern, SL::BYTES, "attr(conv_dgrad_bf16)");
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(Cfg::THREADS), SL::BYTES, s, pa, pb, dx, g.CiP, tiles_m,
                     tiles_n, Kdw / BK);
  return check_hip(hipGetLastError(), "conv_dgrad_bf16 launch");
}

struct WgradPlanB { int tiles_m, tiles_n, nk, splits, ks_per_split, Mtot, KI, bias_parts; int64_t bias_per; };
static WgradPlanB plan_wgrad_bf16(const ConvGeom& g) {
  WgradPlanB p;
  p.KI = 9 * g.CiP;
  p.Mtot = g.B * 2 * g.Hp * 2 * g.Wp;
  p.tiles_m = (p.KI + CfgWb::BM - 1) / CfgWb::BM;
  p.tiles_n = (g.Co + CfgWb::BN - 1) / CfgWb::BN;
  p.nk = (p.Mtot + BKB - 1) / BKB;
  int splits = 512 / (p.tiles_m * p.tiles_n);
  if (splits < 1) splits = 1;
  const int max_splits = p.nk / 8 > 1 ? p.nk / 8 : 1;
  if (splits > max_splits) splits = max_splits;
  p.ks_per_split = (p.nk + splits - 1) / splits;
  p.splits = (p.nk + p.ks_per_split - 1) / p.ks_per_split;
  const int64_t windows = (int64_t)g.B * g.Hp * g.Wp;
  int64_t parts = (windows + 63) / 64;
  if (parts > 2048) parts = 2048;
  p.bias_per = (windows + parts - 1) / parts;
  p.bias_parts = (int)((windows + p.bias_per - 1) / p.bias_per);
  return p;
}

}  // namespace vqa

using namespace vqa;

extern "C" {

/* ---- bf16 path (BASELINE configs[3]) ------------------------------------------------------------------------ */
int vqa_conv_pack_weights_bf16(const float* w, void* wfT, void* wdT, int Co, int Ci, int CiP, vqa_stream_t stream) {
  VQA_REQUIRE(w && wfT && Ci <= CiP && CiP % 8 == 0, "vqa_conv_pack_weights_bf16: bad args Ci=%d CiP=%d", Ci, CiP);
  const int total = 9 * CiP * Co;
  hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                     static_cast<uint16_t*>(wfT), static_cast<uint16_t*>(wdT), Co, Ci, CiP);
  return check_hip(hipGetLastError(), "pack_weights_bf16 launch");
}

static int check_geom_bf16(const char* fn, const ConvGeom& g, int k_channels) {
  int rc = check_geom(fn, g);
  if (rc) return rc;
  VQA_REQUIRE(k_channels % 64 == 0, "%s: the reduction-side channel count (%d) must be a multiple of 64", fn, k_channels);
  VQA_REQUIRE(g.CiP % 8 == 0 && g.Co % 8 == 0, "%s: channel counts must be multiples of 8 (CiP=%d Co=%d)", fn, g.CiP, g.Co);
  return VQA_OK;
}

int vqa_conv3x3_relu_pool_fwd_bf16(const void* x, const void* wfT, const float* bias, void* pooled, int pooled_is_bf16,
                                   uint8_t* argmax, int B, int H, int W, int CiP, int Co, int stride, int tag,
                                   vqa_stream_t stream) {
  VQA_REQUIRE(x && wfT && bias && pooled && argmax && B > 0, "vqa_conv3x3_relu_pool_fwd_bf16: null pointer");
  const int chunk = batch_chunk(B, H, W, CiP, Co, stride);
  VQA_REQUIRE(chunk > 0, "vqa_conv3x3_relu_pool_fwd_bf16: one %dx%dx%d image reaches 4 GiB", H, W, CiP);
  const ConvGeom g1 = make_geom(1, H, W, CiP, Co, stride);
  hipStream_t s = (hipStream_t)stream;
  set_launch_tag(tag);
  ProfScope prof(VQA_K_CONV_FWD, s);
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = B - b0 < chunk ? B - b0 : chunk;
    const ConvGeom g = make_geom(nb, H, W, CiP, Co, stride);
    int rc = check_geom_bf16("vqa_conv3x3_relu_pool_fwd_bf16", g, CiP);
    if (rc) return rc;
    const char* xc = static_cast<const char*>(x) + (int64_t)b0 * H * W * CiP * 2;
    const int64_t po = (int64_t)b0 * g1.Hp * g1.Wp * Co;
    void* pc = static_cast<char*>(pooled) + po * (pooled_is_bf16 ? 2 : 4);
    if (Co > 64 && knobs().big_tiles == 1)
      rc = pooled_is_bf16 ? launch_fwd_bf16<Cfg256x128, true>(xc, wfT, bias, pc, argmax + po, g, s)
                          : launch_fwd_bf16<Cfg256x128, false>(xc, wfT, bias, pc, argmax + po, g, s);
    else if (Co > 64) rc = pooled_is_bf16 ? launch_fwd_bf16<Cfg128F, true>(xc, wfT, bias, pc, argmax + po, g, s)
                                     : launch_fwd_bf16<Cfg128F, false>(xc, wfT, bias, pc, argmax + po, g, s);
    else rc = pooled_is_bf16 ? launch_fwd_bf16<Cfg128x64, true>(xc, wfT, bias, pc, argmax + po, g, s)
                             : launch_fwd_bf16<Cfg128x64, false>(xc, wfT, bias, pc, argmax + po, g, s);
    if (rc) return rc;
  }
  return VQA_OK;
}

int vqa_conv3x3_dgrad_bf16(const void* dpooled, const uint8_t* argmax, const void* wdT, void* dx, int dx_is_bf16, int B,
                           int H, int W, int CiP, int Co, int stride, int tag, vqa_stream_t stream) {
  VQA_REQUIRE(dpooled && argmax && wdT && dx && B > 0, "vqa_conv3x3_dgrad_bf16: null pointer");
  const int chunk = batch_chunk(B, H, W, CiP, Co, stride);
  VQA_REQUIRE(chunk > 0, "vqa_conv3x3_dgrad_bf16: one %dx%dx%d image reaches 4 GiB", H, W, CiP);
  const ConvGeom g1 = make_geom(1, H, W, CiP, Co, stride);
  hipStream_t s = (hipStream_t)stream;
  set_launch_tag(tag);
  ProfScope prof(VQA_K_CONV_DGRAD, s);
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = B - b0 < chunk ? B - b0 : chunk;
    const ConvGeom g = make_geom(nb, H, W, CiP, Co, stride);
    int rc = check_geom_bf16("vqa_conv3x3_dgrad_bf16", g, Co);
    if (rc) return rc;
    const int64_t po = (int64_t)b0 * g1.Hp * g1.Wp * Co, xo = (int64_t)b0 * H * W * CiP;
    const char* dpc = static_cast<const char*>(dpooled) + po * 2;
    void* dxc = static_cast<char*>(dx) + xo * (dx_is_bf16 ? 2 : 4);
    if (knobs().big_tiles == 1)
      rc = CiP > 64 ? (dx_is_bf16 ? launch_dgrad_bf16<Cfg256x128, true>(dpc, argmax + po, wdT, dxc, g, s)
                                  : launch_dgrad_bf16<Cfg256x128, false>(dpc, argmax + po, wdT, dxc, g, s))
                    : (dx_is_bf16 ? launch_dgrad_bf16<Cfg256x64, true>(dpc, argmax + po, wdT, dxc, g, s)
                                  : launch_dgrad_bf16<Cfg256x64, false>(dpc, argmax + po, wdT, dxc, g, s));
    else if (CiP > 64) rc = dx_is_bf16 ? launch_dgrad_bf16<Cfg128, true>(dpc, argmax + po, wdT, dxc, g, s)
                                  : launch_dgrad_bf16<Cfg128, false>(dpc, argmax + po, wdT, dxc, g, s);
    else if (knobs().big_tiles == 0)
      rc = dx_is_bf16 ? launch_dgrad_bf16<Cfg128x64, true>(dpc, argmax + po, wdT, dxc, g, s)
                      : launch_dgrad_bf16<Cfg128x64, false>(dpc, argmax + po, wdT, dxc, g, s);
    else rc = dx_is_bf16 ? launch_dgrad_bf16<Cfg128x64L8, true>(dpc, argmax + po, wdT, dxc, g, s)
                         : launch_dgrad_bf16<Cfg128x64L8, false>(dpc, argmax + po, wdT, dxc, g, s);
    if (rc) return rc;
  }
  return VQA_OK;
}

int64_t vqa_conv3x3_wgrad_bf16_workspace_bytes(int B, int H, int W, int CiP, int Co, int stride) {
  const ConvGeom g1 = make_geom(1, H, W, CiP, Co, stride);
  if (g1.Hp <= 0 || g1.Wp <= 0 || B <= 0) return 0;
  const int chunk = batch_chunk(B, H, W, CiP, Co, stride);
  if (chunk <= 0) return 0;
  int64_t parts = 0, bparts = 0;
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const WgradPlanB p = plan_wgrad_bf16(make_geom(B - b0 < chunk ? B - b0 : chunk, H, W, CiP, Co, stride));
    parts += p.splits;
    bparts += p.bias_parts;
  }
  return (parts * (int64_t)9 * CiP * Co + bparts * Co) * 4;
}

int vqa_conv3x3_wgrad_bf16(const void* x, const void* dpooled, const uint8_t* argmax, float* dw, float* dbias, int B,
                           int H, int W, int CiP, int Ci, int Co, int stride, float* workspace,
                           int64_t workspace_bytes, int tag, vqa_stream_t stream) {
  VQA_REQUIRE(x && dpooled && argmax && dw && dbias && workspace && B > 0, "vqa_conv3x3_wgrad_bf16: null pointer");
  VQA_REQUIRE(Ci >= 1 && Ci <= CiP, "vqa_conv3x3_wgrad_bf16: Ci=%d CiP=%d", Ci, CiP);
  VQA_REQUIRE(Co % 8 == 0 && Co <= 2048, "vqa_conv3x3_wgrad_bf16: Co=%d must be a multiple of 8, at most 2048", Co);
  const int chunk = batch_chunk(B, H, W, CiP, Co, stride);
  VQA_REQUIRE(chunk > 0, "vqa_conv3x3_wgrad_bf16: one %dx%dx%d image reaches 4 GiB", H, W, CiP);
  const int64_t need = vqa_conv3x3_wgrad_bf16_workspace_bytes(B, H, W, CiP, Co, stride);
  if (workspace_bytes < need) {
    set_error("vqa_conv3x3_wgrad_bf16: workspace %lld < %lld", (long long)workspace_bytes, (long long)need);
    return VQA_ERR_WORKSPACE;
  }
  const ConvGeom g1 = make_geom(1, H, W, CiP, Co, stride);
  const int KI = 9 * CiP;
  int parts = 0, bparts = 0;
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const WgradPlanB p = plan_wgrad_bf16(make_geom(B - b0 < chunk ? B - b0 : chunk, H, W, CiP, Co, stride));
    parts += p.splits;
    bparts += p.bias_parts;
  }
  float* const bias_slab0 = workspace + (int64_t)parts * KI * Co;
  hipStream_t s = (hipStream_t)stream;
  set_launch_tag(tag);
  ProfScope prof(VQA_K_CONV_WGRAD, s);
  using SL = SmemLayoutB<CfgWb, false, false>;
  auto kern = conv_wgrad_bf16_kernel<CfgWb>;
  int rc = set_smem(kern, SL::BYTES, "attr(conv_wgrad_bf16)");
  if (rc) return rc;
  int done = 0, bdone = 0;
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = B - b0 < chunk ? B - b0 : chunk;
    const ConvGeom g = make_geom(nb, H, W, CiP, Co, stride);
    rc = check_geom_bf16("vqa_conv3x3_wgrad_bf16", g, 64);
    if (rc) return rc;
    const WgradPlanB p = plan_wgrad_bf16(g);
    WgradGeom wg{g.H, g.W, g.CiP, g.Hp, g.Wp, g.Co, g.stride, p.Mtot};
    const int64_t po = (int64_t)b0 * g1.Hp * g1.Wp * Co;
    const char* xc = static_cast<const char*>(x) + (int64_t)b0 * H * W * CiP * 2;
    const char* dpc = static_cast<const char*>(dpooled) + po * 2;
    typename WgradAb<CfgWb::BM, CfgWb::LT>::Params pa{xc, wg, p.KI};
    typename WgradBb<CfgWb::BN, CfgWb::LT>::Params pb{dpc, argmax + po, wg};
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n * p.splits), dim3(CfgWb::THREADS), SL::BYTES, s, pa, pb,
                       workspace + (int64_t)done * KI * Co, p.tiles_m, p.tiles_n, p.nk, p.ks_per_split);
    rc = check_hip(hipGetLastError(), "conv_wgrad_bf16 launch");
    if (rc) return rc;
    hipLaunchKernelGGL(conv_bias_grad_bf16_kernel, dim3(p.bias_parts), dim3(256), (size_t)(256 / (Co / 8)) * Co * 4, s,
                       reinterpret_cast<const uint16_t*>(dpc), argmax + po, bias_slab0 + (int64_t)bdone * Co,
                       (int64_t)g.B * g.Hp * g.Wp, Co, p.bias_per);
    rc = check_hip(hipGetLastError(), "conv_bias_grad_bf16 launch");
    if (rc) return rc;
    done += p.splits;
    bdone += p.bias_parts;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((KI * Co + 63) / 64), dim3(256), 0, s, workspace, dw, parts, KI, CiP, Ci, Co);
  rc = check_hip(hipGetLastError(), "wgrad_reduce launch");
  if (rc) return rc;
  hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3((Co + 31) / 32), dim3(256), 0, s, bias_slab0, dbias, bparts, Co);
  return check_hip(hipGetLastError(), "wgrad_bias_reduce launch");
}

}  // extern "C"
