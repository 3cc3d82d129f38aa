// Image encoder kernels: Conv2d(3x3, stride, pad 0) + ReLU + MaxPool2d(2,2), forward / dgrad / wgrad,
// as implicit GEMMs on the fp32 MFMA engine (gemm_core.hpp).
//
// Reference: models/model.py:72-84 (ImageNet2).  Layout: NHWC, channels padded to a multiple of 4.
//
// forward   M = 4 * (#pool windows): row m = 4*w + j is conv-output pixel j = dy*2+dx of pool window
//           w = (b, py, px).  In the MFMA C layout a lane holds rows (r&3) + 8*(r>>2) + 4*(lane>>5)
//           of one column, i.e. the four pixels of a window sit in four consecutive accumulator
//           registers of ONE lane: bias + ReLU + 2x2 max + arg-max are register-only, and the
//           pre-pool activation (3.2 GB for conv0 at B = 256) never exists in memory.
//           K = 9*CiP ordered (ky, kx, ci): the A "row" of a pixel for one tap is CiP contiguous floats.
// dgrad     rows = conv-INPUT pixels, K = (ky, kx, co); the A loader rebuilds dY on the fly from the
//           pooled gradient and the stored arg-max byte (one non-zero per 2x2 window).
// wgrad     out[(ky,kx,ci)][co] = sum over conv-output pixels; split along that (huge) reduction over
//           workgroups into slabs, then reduced and transposed to the torch layout [Co][Ci][3][3].
#include "bf16_core.hpp"

namespace vqa {

int colsum_launch(const float* x, int64_t ld, const uint8_t* mask, int64_t rows, int cols, float* out,
                  int accumulate, float* ws, int64_t ws_bytes, hipStream_t s);
int64_t colsum_ws_bytes(int64_t rows, int cols);

#include "conv_device.inc"
#include "conv_host.inc"

using Cfg128 = TileCfg<128, 128, 2, 2>;
using Cfg128x64 = TileCfg<128, 64, 2, 2>;
// 256-row tiles for the pixel-major GEMMs (forward / dgrad: M = millions of pixels): 25 % fewer bytes
// staged per MFMA than 128x128 (tools/mfma_ws.hip: 83 % -> 88 % of the MFMA rate in the same skeleton).
#ifndef VQA_BIG_PF
#define VQA_BIG_PF 0
#endif
using Cfg256x128 = TileCfg<256, 128, 4, 2, 4, VQA_BIG_PF>;   // 8 MFMA waves (64x64 each) + 4 loader waves, one workgroup per CU
using Cfg256x64 = TileCfg<256, 64, 4, 2, 4, VQA_BIG_PF>;     // 8 MFMA waves (64x32 each)
using Cfg64 = TileCfg<64, 64, 2, 2>;
// wgrad rows are (tap, ci): 9*CiP = 576 for CiP = 64 is 4.5 x 128 but 6 x 96 (1 x 4 MFMA waves of 96x32)
#ifndef VQA_WGRAD_PF
#define VQA_WGRAD_PF 3
#endif
using Cfg96x128 = TileCfg<96, 128, 1, 4, 4, VQA_WGRAD_PF>;
// wgrad streams new pixels from HBM every K-step: three K-steps of loads in flight (75 -> 78 % on conv1 in spite
// of ~55 spilled loader registers; four spill 100+ and lose 15 points; dgrad's 128x64 tile loses 9 points at three)
using Cfg128W = TileCfg<128, 128, 2, 2, 4, VQA_WGRAD_PF>;
#ifndef VQA_W192_PF
#define VQA_W192_PF 3
#endif
// 192 rows, 8 MFMA waves (2 x 4, 96x32 each) + 4 loader waves, one workgroup per CU (84 KB LDS, 164 VGPRs): the routed
// B operand is reloaded by 3 (conv1) / 6 (conv2) row tiles instead of 6 / 9 (see plan_wgrad for when it is used)
using Cfg192x128W = TileCfg<192, 128, 2, 4, 4, VQA_W192_PF>;
using Cfg384x128W = TileCfg<384, 128, 4, 2, 4, 2>;   // 9*CiP = 1152 as three row tiles (see plan_wgrad)
// the same tiles with 8 loader waves (1024 threads): two loader waves per SIMD for loader-bound K loops
using Cfg256x128L8 = TileCfg<256, 128, 4, 2, 8>;
using Cfg256x64L8 = TileCfg<256, 64, 4, 2, 8>;
// dgrad with CiP <= 64 (conv1): a K-step is only 32 MFMAs per wave, so the loaders are the long pole; with 8
// loader waves (768 threads, 74 VGPRs: two workgroups = 6 waves per SIMD) 70.6 -> 73.0 % on the same box
#ifndef VQA_DGRAD_PF
#define VQA_DGRAD_PF 2
#endif
using Cfg128x64L8 = TileCfg<128, 64, 2, 2, 8, VQA_DGRAD_PF>;

template <class Cfg, bool U>
static int launch_fwd(const float* x, const float* wf, const float* bias, float* pooled, uint8_t* amax,
                      const ConvGeom& g, hipStream_t s) {
  using SL = SmemLayout<Cfg, true, false>;
  const int nWin = g.B * g.Hp * g.Wp, K = 9 * g.CiP;
  typename ConvFwdA<Cfg::NVA, Cfg::LT, U>::Params pa{x, g.H, g.W, g.CiP, g.Hp, g.Wp, g.stride, nWin, K};
  typename PlainC<Cfg::NVB, Cfg::LT>::Params pb{wf, g.Co, g.Co, K};
  const int tiles_m = (4 * nWin + Cfg::BM - 1) / Cfg::BM, tiles_n = (g.Co + Cfg::BN - 1) / Cfg::BN;
  const int slots = 256 * SL::WG_PER_CU, tiles = tiles_m * tiles_n;
  // Persistent tiles measured neutral to slower on the conv kernels (conv1 fwd 3.83 -> 3.81 ms, conv2 dgrad
  // 3.65 -> 3.93 ms: their K loops are long and the second workgroup / MFMA wave of the SIMD already covers a
  // tile's prologue and epilogue), so they are opt-in (VQA_PERSISTENT=1) and parity-tested that way.
  const bool persistent = knobs().persistent == 1;
  if (persistent) {
    auto pk = conv_fwd_persistent_kernel<Cfg, U>;
    { int rc = set_smem(pk, SL::BYTES, "attr(conv_fwd_p)"); if (rc) return rc; }
    hipLaunchKernelGGL(pk, dim3(tiles < slots ? tiles : slots), dim3(Cfg::THREADS), SL::BYTES, s, pa, pb, bias,
                       pooled, amax, g.Co, tiles_m, tiles_n, (K + BK - 1) / BK);
    return check_hip(hipGetLastError(), "conv_fwd_persistent launch");
  }
  auto kern = conv_fwd_kernel<Cfg, U>;
  { int rc = set_smem(kern, SL::BYTES, "attr(conv_fwd)"); if (rc) return rc; }
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(Cfg::THREADS), SL::BYTES, s, pa, pb, bias, pooled, amax,
                     g.Co, tiles_m, tiles_n, (K + BK - 1) / BK);
  return check_hip(hipGetLastError(), "conv_fwd launch");
}

template <class Cfg, bool U>
static int launch_dgrad(const float* dp, const uint8_t* am, const float* wd, float* dx, const ConvGeom& g,
                        hipStream_t s) {
  using SL = SmemLayout<Cfg, true, false>;
  const int rows = g.B * g.H * g.W, K = 9 * g.Co;
  typename ConvDgradA<Cfg::NVA, Cfg::LT, U>::Params pa{dp, am, g.H, g.W, g.Hp, g.Wp, g.Co, g.stride, rows, K};
  typename PlainC<Cfg::NVB, Cfg::LT>::Params pb{wd, g.CiP, g.CiP, K};
  const int tiles_m = (rows + Cfg::BM - 1) / Cfg::BM, tiles_n = (g.CiP + Cfg::BN - 1) / Cfg::BN;
  const int slots = 256 * SL::WG_PER_CU, tiles = tiles_m * tiles_n;
  const bool persistent = knobs().persistent == 1;   // see launch_fwd
  if (persistent) {
    auto pk = conv_dgrad_persistent_kernel<Cfg, U>;
    { int rc = set_smem(pk, SL::BYTES, "attr(conv_dgrad_p)"); if (rc) return rc; }
    hipLaunchKernelGGL(pk, dim3(tiles < slots ? tiles : slots), dim3(Cfg::THREADS), SL::BYTES, s, pa, pb, dx,
                       g.CiP, tiles_m, tiles_n, (K + BK - 1) / BK);
    return check_hip(hipGetLastError(), "conv_dgrad_persistent launch");
  }
  auto kern = conv_dgrad_kernel<Cfg, U>;
  { int rc = set_smem(kern, SL::BYTES, "attr(conv_dgrad)"); if (rc) return rc; }
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(Cfg::THREADS), SL::BYTES, s, pa, pb, dx, g.CiP, tiles_m,
                     tiles_n, (K + BK - 1) / BK);
  return check_hip(hipGetLastError(), "conv_dgrad launch");
}

struct WgradPlan { int big, bm, tiles_m, tiles_n, nk, splits, ks_per_split, Mtot, KI; };
static WgradPlan plan_wgrad(const ConvGeom& g) {
  WgradPlan p;
  p.KI = 9 * g.CiP;
  p.Mtot = g.B * 2 * g.Hp * 2 * g.Wp;
  p.big = (p.KI >= 128 && g.Co >= 128) ? 1 : 0;
  const int bn = p.big ? 128 : 64;
  // 96-row tiles when they waste fewer rows than 128-row tiles (KI = 576: 576 vs 640 rows of MFMA work)
  p.bm = !p.big ? 64 : ((p.KI + 95) / 96 * 96 < (p.KI + 127) / 128 * 128 ? 96 : 128);
  // 9*CiP = 576 (CiP = 64): three 192-row tiles (8 MFMA waves, one workgroup per CU) instead of six 96-row tiles halve
  // the reloads of the routed B operand: 76.6 -> 80.3 % with the early K-step barrier (before it: +1 point); where
  // 128-row tiles fit exactly (9*CiP = 1152) two 128x128 workgroups per CU stay ahead (83.0 vs 79.4 %).
  // VQA_WGRAD_192=0 / 1 disables / forces it wherever 9*CiP % 192 == 0.
  {
    const int w192 = knobs().wgrad_192, w384 = knobs().wgrad_384;
    const bool can = p.big && p.KI % 192 == 0 && g.CiP % BK == 0 && g.Co % BK == 0 && 2 * g.Wp >= BK;
    if (can && (w192 >= 0 ? w192 == 1 : p.bm == 96)) p.bm = 192;
    // 9*CiP = 1152 (CiP = 128): three 384-row tiles (8 MFMA waves of 96x64, one workgroup per CU, 133 KB LDS) reload
    // the routed B operand 3x instead of 9x: 82.4 -> 86.6 % (same box).  VQA_WGRAD_384=0 / 1 disables / forces it.
    if (can && p.KI % 384 == 0 && w192 != 1 && (w384 >= 0 ? w384 == 1 : true)) p.bm = 384;
  }
  p.tiles_m = (p.KI + p.bm - 1) / p.bm;
  p.tiles_n = (g.Co + bn - 1) / bn;
  p.nk = (p.Mtot + BK - 1) / BK;
  const int tiles = p.tiles_m * p.tiles_n;
  // 2 workgroups fit a CU (LDS): tiles * splits must not exceed the 512 resident slots, or the few
  // workgroups left over run alone in a second round and double the kernel's time.
  int splits = (p.bm >= 192 ? 256 : 512) / tiles;
  if (splits < 1) splits = 1;
  const int max_splits = p.nk / 8 > 1 ? p.nk / 8 : 1;
  if (splits > max_splits) splits = max_splits;
  p.ks_per_split = (p.nk + splits - 1) / splits;
  p.splits = (p.nk + p.ks_per_split - 1) / p.ks_per_split;
  return p;
}

template <class Cfg, bool U>
static int launch_wgrad(const float* x, const float* dp, const uint8_t* am, float* slab, float* bias_slab,
                        const ConvGeom& g, const WgradPlan& p, hipStream_t s) {
  using SL = SmemLayout<Cfg, false, false>;
  WgradGeom wg{g.H, g.W, g.CiP, g.Hp, g.Wp, g.Co, g.stride, p.Mtot};
  typename WgradA<Cfg::NVA, Cfg::LT, U>::Params pa{x, wg, p.KI};
  typename WgradB<Cfg::NVB, Cfg::LT, U>::Params pb{dp, am, wg};
  auto kern = conv_wgrad_kernel<Cfg, U>;
  { int rc = set_smem(kern, SL::BYTES, "attr(conv_wgrad)"); if (rc) return rc; }
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n * p.splits), dim3(Cfg::THREADS), SL::BYTES, s, pa, pb, slab,
                     bias_slab, p.tiles_m, p.tiles_n, p.nk, p.ks_per_split);
  return check_hip(hipGetLastError(), "conv_wgrad launch");
}

}  // namespace vqa

using namespace vqa;

extern "C" {

int vqa_nchw_to_nhwc4(const float* x, float* y, int B, int C, int H, int W, vqa_stream_t stream) {
  VQA_REQUIRE(x && y && C >= 1 && C <= 4, "vqa_nchw_to_nhwc4: C=%d must be 1..4", C);
  const int64_t HW = (int64_t)H * W, total = (int64_t)B * HW;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, C, HW, total);
  return check_hip(hipGetLastError(), "nchw_to_nhwc4 launch");
}

int vqa_conv_pack_weights(const float* w, float* wf, float* wd, int Co, int Ci, int CiP, vqa_stream_t stream) {
  VQA_REQUIRE(w && wf && Ci <= CiP && CiP % 4 == 0, "vqa_conv_pack_weights: bad args Ci=%d CiP=%d", Ci, CiP);
  const int total = 9 * CiP * Co;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, wf, wd,
                     Co, Ci, CiP);
  return check_hip(hipGetLastError(), "pack_weights launch");
}

static int fwd_chunk(const float* x, const float* wf, const float* bias, float* pooled, uint8_t* argmax, int B, int H,
                     int W, int CiP, int Co, int stride, vqa_stream_t stream) {
  const ConvGeom g = make_geom(B, H, W, CiP, Co, stride);
  int rc = check_geom("vqa_conv3x3_relu_pool_fwd", g);
  if (rc) return rc;
  // Two 128x128 workgroups per CU beat one 256x128 workgroup (8 MFMA + 4 or 8 loader waves) since the loaders
  // left the VALU: conv1 / conv2 forward 84.4 / 88.3 % against 82.7 / 84.7 % (4 loader waves) and 83.6 / 85.2 %
  // (8); on dgrad the gap is wider (72 / 81 % against 56 / 70 % and 66 / 75 %).  The 256-row kernels stay
  // selectable with VQA_BIG_TILES=1 (4 loader waves) / 3 (8) and are parity-tested that way.
  const int bt = knobs().big_tiles;
  const bool many_rows = bt == 1;
  // channel counts that are not multiples of BK take the general per-lane-tap loaders (one tile shape)
  if (CiP % BK != 0) return launch_fwd<Cfg128x64, false>(x, wf, bias, pooled, argmax, g, (hipStream_t)stream);
  if (bt == 3 && Co > 64) return launch_fwd<Cfg256x128L8, true>(x, wf, bias, pooled, argmax, g, (hipStream_t)stream);
  if (Co > 64) return many_rows ? launch_fwd<Cfg256x128, true>(x, wf, bias, pooled, argmax, g, (hipStream_t)stream)
                                : launch_fwd<Cfg128, true>(x, wf, bias, pooled, argmax, g, (hipStream_t)stream);
  return launch_fwd<Cfg128x64, true>(x, wf, bias, pooled, argmax, g, (hipStream_t)stream);
}

int vqa_conv3x3_relu_pool_fwd(const float* x, const float* wf, const float* bias, float* pooled, uint8_t* argmax,
                              int B, int H, int W, int CiP, int Co, int stride, int tag, vqa_stream_t stream) {
  VQA_REQUIRE(x && wf && bias && pooled && argmax && B > 0, "vqa_conv3x3_relu_pool_fwd: null pointer");
  const int chunk = batch_chunk(B, H, W, CiP, Co, stride);
  VQA_REQUIRE(chunk > 0, "vqa_conv3x3_relu_pool_fwd: one %dx%dx%d image reaches 4 GiB", H, W, CiP);
  const ConvGeom g1 = make_geom(1, H, W, CiP, Co, stride);
  set_launch_tag(tag);
  ProfScope prof(VQA_K_CONV_FWD, (hipStream_t)stream);
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = B - b0 < chunk ? B - b0 : chunk;
    const int64_t xo = (int64_t)b0 * H * W * CiP, po = (int64_t)b0 * g1.Hp * g1.Wp * Co;
    int rc = fwd_chunk(x + xo, wf, bias, pooled + po, argmax + po, nb, H, W, CiP, Co, stride, stream);
    if (rc) return rc;
  }
  return VQA_OK;
}

static int dgrad_chunk(const float* dpooled, const uint8_t* argmax, const float* wd, float* dx, int B, int H, int W,
                       int CiP, int Co, int stride, vqa_stream_t stream) {
  const ConvGeom g = make_geom(B, H, W, CiP, Co, stride);
  int rc = check_geom("vqa_conv3x3_dgrad", g);
  if (rc) return rc;
  const int bt = knobs().big_tiles;
  const bool many_rows = bt == 1;   // opt-in only, see vqa_conv3x3_relu_pool_fwd
  if (Co % BK != 0) return launch_dgrad<Cfg128x64, false>(dpooled, argmax, wd, dx, g, (hipStream_t)stream);
  if (bt == 3)
    return CiP > 64 ? launch_dgrad<Cfg256x128L8, true>(dpooled, argmax, wd, dx, g, (hipStream_t)stream)
                    : launch_dgrad<Cfg256x64L8, true>(dpooled, argmax, wd, dx, g, (hipStream_t)stream);
  if (CiP > 64) return many_rows ? launch_dgrad<Cfg256x128, true>(dpooled, argmax, wd, dx, g, (hipStream_t)stream)
                                 : launch_dgrad<Cfg128, true>(dpooled, argmax, wd, dx, g, (hipStream_t)stream);
  if (bt == 0) return launch_dgrad<Cfg128x64, true>(dpooled, argmax, wd, dx, g, (hipStream_t)stream);
  return many_rows ? launch_dgrad<Cfg256x64, true>(dpooled, argmax, wd, dx, g, (hipStream_t)stream)
                   : launch_dgrad<Cfg128x64L8, true>(dpooled, argmax, wd, dx, g, (hipStream_t)stream);
}

int vqa_conv3x3_dgrad(const float* dpooled, const uint8_t* argmax, const float* wd, float* dx, int B, int H, int W,
                      int CiP, int Co, int stride, int tag, vqa_stream_t stream) {
  VQA_REQUIRE(dpooled && argmax && wd && dx && B > 0, "vqa_conv3x3_dgrad: null pointer");
  const int chunk = batch_chunk(B, H, W, CiP, Co, stride);
  VQA_REQUIRE(chunk > 0, "vqa_conv3x3_dgrad: one %dx%dx%d image reaches 4 GiB", H, W, CiP);
  const ConvGeom g1 = make_geom(1, H, W, CiP, Co, stride);
  set_launch_tag(tag);
  ProfScope prof(VQA_K_CONV_DGRAD, (hipStream_t)stream);
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = B - b0 < chunk ? B - b0 : chunk;
    const int64_t xo = (int64_t)b0 * H * W * CiP, po = (int64_t)b0 * g1.Hp * g1.Wp * Co;
    int rc = dgrad_chunk(dpooled + po, argmax + po, wd, dx + xo, nb, H, W, CiP, Co, stride, stream);
    if (rc) return rc;
  }
  return VQA_OK;
}

int64_t vqa_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int CiP, int Co, int stride) {
  const ConvGeom g1 = make_geom(1, H, W, CiP, Co, stride);
  if (g1.Hp <= 0 || g1.Wp <= 0 || B <= 0) return 0;
  const int chunk = batch_chunk(B, H, W, CiP, Co, stride);
  if (chunk <= 0) return 0;
  int64_t parts = 0;
  for (int b0 = 0; b0 < B; b0 += chunk)
    parts += plan_wgrad(make_geom(B - b0 < chunk ? B - b0 : chunk, H, W, CiP, Co, stride)).splits;
  // one [9*CiP][Co] slab and one partial bias row per split of every chunk
  return parts * ((int64_t)9 * CiP * Co + Co) * 4;
}

int vqa_conv3x3_wgrad(const float* x, const float* dpooled, const uint8_t* argmax, float* dw, float* dbias, int B,
                      int H, int W, int CiP, int Ci, int Co, int stride, float* workspace, int64_t workspace_bytes,
                      int tag, vqa_stream_t stream) {
  VQA_REQUIRE(x && dpooled && argmax && dw && dbias && workspace && B > 0, "vqa_conv3x3_wgrad: null pointer");
  VQA_REQUIRE(Ci >= 1 && Ci <= CiP, "vqa_conv3x3_wgrad: Ci=%d CiP=%d", Ci, CiP);
  const int chunk = batch_chunk(B, H, W, CiP, Co, stride);
  VQA_REQUIRE(chunk > 0, "vqa_conv3x3_wgrad: one %dx%dx%d image reaches 4 GiB", H, W, CiP);
  const int64_t need = vqa_conv3x3_wgrad_workspace_bytes(B, H, W, CiP, Co, stride);
  if (workspace_bytes < need) {
    set_error("vqa_conv3x3_wgrad: workspace %lld < %lld", (long long)workspace_bytes, (long long)need);
    return VQA_ERR_WORKSPACE;
  }
  const ConvGeom g1 = make_geom(1, H, W, CiP, Co, stride);
  const int KI = 9 * CiP;
  int parts = 0;
  for (int b0 = 0; b0 < B; b0 += chunk)
    parts += plan_wgrad(make_geom(B - b0 < chunk ? B - b0 : chunk, H, W, CiP, Co, stride)).splits;
  float* const bias_slab0 = workspace + (int64_t)parts * KI * Co;
  hipStream_t s = (hipStream_t)stream;
  set_launch_tag(tag);
  int rc;
  {
    ProfScope prof(VQA_K_CONV_WGRAD, s);
    int done = 0;
    for (int b0 = 0; b0 < B; b0 += chunk) {
      const int nb = B - b0 < chunk ? B - b0 : chunk;
      const ConvGeom g = make_geom(nb, H, W, CiP, Co, stride);
      rc = check_geom("vqa_conv3x3_wgrad", g);
      if (rc) return rc;
      const WgradPlan p = plan_wgrad(g);
      const float* xc = x + (int64_t)b0 * H * W * CiP;
      const int64_t po = (int64_t)b0 * g1.Hp * g1.Wp * Co;
      float* slab = workspace + (int64_t)done * KI * Co;
      float* bias_slab = bias_slab0 + (int64_t)done * Co;
      const bool uni = CiP % BK == 0 && Co % BK == 0 && 2 * g.Wp >= BK;
      // (8 loader waves on the 192-row tile: +0.5 %, with 3 or 4 K-steps in flight alike; not worth the kernels)
      rc = p.bm == 384 ? launch_wgrad<Cfg384x128W, true>(xc, dpooled + po, argmax + po, slab, bias_slab, g, p, s)
         : p.bm == 192 ? launch_wgrad<Cfg192x128W, true>(xc, dpooled + po, argmax + po, slab, bias_slab, g, p, s)
         : p.bm == 96 ? (uni ? launch_wgrad<Cfg96x128, true>(xc, dpooled + po, argmax + po, slab, bias_slab, g, p, s)
                             : launch_wgrad<Cfg96x128, false>(xc, dpooled + po, argmax + po, slab, bias_slab, g, p, s))
         : p.big ? (uni ? launch_wgrad<Cfg128W, true>(xc, dpooled + po, argmax + po, slab, bias_slab, g, p, s)
                        : launch_wgrad<Cfg128W, false>(xc, dpooled + po, argmax + po, slab, bias_slab, g, p, s))
                 : (uni ? launch_wgrad<Cfg64, true>(xc, dpooled + po, argmax + po, slab, bias_slab, g, p, s)
                        : launch_wgrad<Cfg64, false>(xc, dpooled + po, argmax + po, slab, bias_slab, g, p, s));
      if (rc) return rc;
      done += p.splits;
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((KI * Co + 63) / 64), dim3(256), 0, s, workspace, dw, parts, KI, CiP,
                       Ci, Co);
    rc = check_hip(hipGetLastError(), "wgrad_reduce launch");
    if (rc) return rc;
    hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3((Co + 31) / 32), dim3(256), 0, s, bias_slab0, dbias, parts, Co);
    rc = check_hip(hipGetLastError(), "wgrad_bias_reduce launch");
  }
  return rc;
}

}  // extern "C"

#ifdef VQA_DIAG
// diagnostic build only: read-and-reset the s_memtime sums of the conv kernels (this translation unit)
extern "C" int vqa_diag_read(unsigned long long* out8) {
  unsigned long long z[4] = {0, 0, 0, 0};
  hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(vqa::vqa_diag_buf), sizeof(z)) != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out8 + 4, HIP_SYMBOL(vqa::vqa_diag_ld), sizeof(z)) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(vqa::vqa_diag_ld), z, sizeof(z)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(vqa::vqa_diag_buf), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif
