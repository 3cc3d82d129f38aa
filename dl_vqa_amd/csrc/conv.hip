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
#include "gemm_core.hpp"

namespace vqa {

int colsum_launch(const float* x, int64_t ld, const uint8_t* mask, int64_t rows, int cols, float* out,
                  int accumulate, float* ws, int64_t ws_bytes, hipStream_t s);
int64_t colsum_ws_bytes(int64_t rows, int cols);

struct ConvGeom {
  int B, H, W, CiP, Co, stride, Ho, Wo, Hp, Wp;
};
static ConvGeom make_geom(int B, int H, int W, int CiP, int Co, int stride) {
  ConvGeom g{B, H, W, CiP, Co, stride, 0, 0, 0, 0};
  g.Ho = (H - 3) / stride + 1;
  g.Wo = (W - 3) / stride + 1;
  g.Hp = g.Ho / 2;
  g.Wp = g.Wo / 2;
  return g;
}

// ------------------------------------------------------------------ forward A loader (type R)
template <int NV>
struct ConvFwdA {
  struct Params { const float* x; int H, W, CiP, Hp, Wp, stride, nWin, K; };
  static constexpr bool kTypeR = true;
  const float* rowp[NV];
  bool ok[NV];
  int W, CiP, K, c4;
  __device__ __forceinline__ void init(const Params& q, int row0, int tid) {
    W = q.W; CiP = q.CiP; K = q.K; c4 = 4 * (tid & 7);
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int m = row0 + (tid >> 3) + 32 * p;
      int wl = m >> 2;
      const int j = m & 3;
      ok[p] = wl < q.nWin;
      if (!ok[p]) wl = 0;
      const int px = wl % q.Wp;
      const int t = wl / q.Wp;
      const int py = t % q.Hp;
      const int b = t / q.Hp;
      const int y = (2 * py + (j >> 1)) * q.stride, x = (2 * px + (j & 1)) * q.stride;
      rowp[p] = q.x + ((int64_t)(b * q.H + y) * q.W + x) * q.CiP;
    }
  }
  __device__ __forceinline__ void load(int ks, float4 (&r)[NV]) const {
    const int kk = ks * BK + c4;
    const bool kok = kk < K;
    const int tap = kok ? kk / CiP : 0;
    const int ci = kk - tap * CiP;
    const int ky = tap / 3, kx = tap - 3 * ky;
    const int off = (ky * W + kx) * CiP + ci;
#pragma unroll
    for (int p = 0; p < NV; ++p)
      r[p] = (kok && ok[p]) ? *reinterpret_cast<const float4*>(rowp[p] + off) : f4zero();
  }
};

template <class Cfg>
__global__ __launch_bounds__(256) void conv_fwd_kernel(typename ConvFwdA<Cfg::NVA>::Params pa,
                                                       typename PlainC<Cfg::NVB>::Params pb,
                                                       const float* __restrict__ bias, float* pooled,
                                                       uint8_t* amax, int Co, int tiles_n, int nk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const TileCoord tc = tile_coord(tiles_n);
  const int m0 = tc.mt * Cfg::BM, n0 = tc.nt * Cfg::BN;
  ConvFwdA<Cfg::NVA> al; al.init(pa, m0, tid);
  PlainC<Cfg::NVB> bl; bl.init(pb, n0, tid);
  f32x16 acc[Cfg::TM][Cfg::TN];
  acc_zero<Cfg>(acc);
  gemm_mainloop<Cfg>(al, bl, acc, 0, nk, pa.K, smem);

  const int h = lane >> 5;
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const int col = n0 + acc_col<Cfg>(wn, j, lane);
    const float bv = col < Co ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int wl = (m0 + wm * Cfg::WM + 32 * i + 8 * g + 4 * h) >> 2;
        float best = acc[i][j][4 * g];
        int a = 0;
        if (acc[i][j][4 * g + 1] > best) { best = acc[i][j][4 * g + 1]; a = 1; }
        if (acc[i][j][4 * g + 2] > best) { best = acc[i][j][4 * g + 2]; a = 2; }
        if (acc[i][j][4 * g + 3] > best) { best = acc[i][j][4 * g + 3]; a = 3; }
        best += bv;
        if (wl < pa.nWin && col < Co) {
          const int64_t o = (int64_t)wl * Co + col;
          pooled[o] = best > 0.f ? best : 0.f;
          amax[o] = best > 0.f ? (uint8_t)a : (uint8_t)4;
        }
      }
  }
}

// ------------------------------------------------------------------ pooled-gradient expansion
__device__ __forceinline__ float4 route4(const float* dp, const uint8_t* am, int64_t off, int j) {
  const float4 d = *reinterpret_cast<const float4*>(dp + off);
  const uchar4 id = *reinterpret_cast<const uchar4*>(am + off);
  float4 r;
  r.x = id.x == j ? d.x : 0.f;
  r.y = id.y == j ? d.y : 0.f;
  r.z = id.z == j ? d.z : 0.f;
  r.w = id.w == j ? d.w : 0.f;
  return r;
}

// ------------------------------------------------------------------ dgrad A loader (type R)
template <int NV>
struct ConvDgradA {
  struct Params { const float* dp; const uint8_t* am; int H, W, Hp, Wp, Co, stride, rows, K; };
  static constexpr bool kTypeR = true;
  Params q;
  int b[NV], y[NV], x[NV];
  bool ok[NV];
  int c4;
  __device__ __forceinline__ void init(const Params& q_, int row0, int tid) {
    q = q_; c4 = 4 * (tid & 7);
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      int m = row0 + (tid >> 3) + 32 * p;
      ok[p] = m < q.rows;
      if (!ok[p]) m = 0;
      x[p] = m % q.W;
      const int t = m / q.W;
      y[p] = t % q.H;
      b[p] = t / q.H;
    }
  }
  __device__ __forceinline__ void load(int ks, float4 (&r)[NV]) const {
    const int kk = ks * BK + c4;
    const bool kok = kk < q.K;
    const int tap = kok ? kk / q.Co : 0;
    const int co = kk - tap * q.Co;
    const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      int yy = y[p] - ky, xx = x[p] - kx;
      bool v = kok && ok[p] && yy >= 0 && xx >= 0;
      if (q.stride == 2) { v = v && !(yy & 1) && !(xx & 1); yy >>= 1; xx >>= 1; }
      v = v && yy < 2 * q.Hp && xx < 2 * q.Wp;
      if (v) {
        const int j = ((yy & 1) << 1) | (xx & 1);
        const int64_t off = ((int64_t)(b[p] * q.Hp + (yy >> 1)) * q.Wp + (xx >> 1)) * q.Co + co;
        r[p] = route4(q.dp, q.am, off, j);
      } else {
        r[p] = f4zero();
      }
    }
  }
};

template <class Cfg>
__global__ __launch_bounds__(256) void conv_dgrad_kernel(typename ConvDgradA<Cfg::NVA>::Params pa,
                                                         typename PlainC<Cfg::NVB>::Params pb, float* dx,
                                                         int CiP, int tiles_n, int nk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const TileCoord tc = tile_coord(tiles_n);
  const int m0 = tc.mt * Cfg::BM, n0 = tc.nt * Cfg::BN;
  ConvDgradA<Cfg::NVA> al; al.init(pa, m0, tid);
  PlainC<Cfg::NVB> bl; bl.init(pb, n0, tid);
  f32x16 acc[Cfg::TM][Cfg::TN];
  acc_zero<Cfg>(acc);
  gemm_mainloop<Cfg>(al, bl, acc, 0, nk, pa.K, smem);
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const int col = n0 + acc_col<Cfg>(wn, j, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + acc_row<Cfg>(wm, i, r, lane);
        if (row < pa.rows && col < CiP) dx[(int64_t)row * CiP + col] = acc[i][j][r];
      }
    }
}

// ------------------------------------------------------------------ wgrad loaders (type C)
struct WgradGeom { int H, W, CiP, Hp, Wp, Co, stride, Mtot; };

template <int NV>
struct WgradA {  // A(i = (ky,kx,ci), m) = x[b][yo*s+ky][xo*s+kx][ci]
  struct Params { const float* x; WgradGeom g; int KI; };
  static constexpr bool kTypeR = false;
  Params q;
  int ioff[NV];
  bool iok[NV];
  int kr;
  __device__ __forceinline__ void init(const Params& q_, int i0, int tid) {
    q = q_; kr = tid >> 3;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int i = i0 + 4 * ((tid & 7) + 8 * p);
      iok[p] = i < q.KI;
      const int tap = iok[p] ? i / q.g.CiP : 0;
      const int ci = iok[p] ? i - tap * q.g.CiP : 0;
      const int ky = tap / 3, kx = tap - 3 * ky;
      ioff[p] = (ky * q.g.W + kx) * q.g.CiP + ci;
    }
  }
  __device__ __forceinline__ void load(int ks, float4 (&r)[NV]) const {
    const int m = ks * BK + kr;
    const bool mok = m < q.g.Mtot;
    const int Wo2 = 2 * q.g.Wp, Ho2 = 2 * q.g.Hp;
    const int mm = mok ? m : 0;
    const int xo = mm % Wo2;
    const int t = mm / Wo2;
    const int yo = t % Ho2;
    const int b = t / Ho2;
    const float* base = q.x + ((int64_t)(b * q.g.H + yo * q.g.stride) * q.g.W + xo * q.g.stride) * q.g.CiP;
#pragma unroll
    for (int p = 0; p < NV; ++p)
      r[p] = (mok && iok[p]) ? *reinterpret_cast<const float4*>(base + ioff[p]) : f4zero();
  }
};

template <int NV>
struct WgradB {  // B(m, co) = dY routed from the pooled gradient
  struct Params { const float* dp; const uint8_t* am; WgradGeom g; };
  static constexpr bool kTypeR = false;
  Params q;
  int co[NV];
  int kr;
  __device__ __forceinline__ void init(const Params& q_, int n0, int tid) {
    q = q_; kr = tid >> 3;
#pragma unroll
    for (int p = 0; p < NV; ++p) co[p] = n0 + 4 * ((tid & 7) + 8 * p);
  }
  __device__ __forceinline__ void load(int ks, float4 (&r)[NV]) const {
    const int m = ks * BK + kr;
    const bool mok = m < q.g.Mtot;
    const int Wo2 = 2 * q.g.Wp, Ho2 = 2 * q.g.Hp;
    const int mm = mok ? m : 0;
    const int xo = mm % Wo2;
    const int t = mm / Wo2;
    const int yo = t % Ho2;
    const int b = t / Ho2;
    const int j = ((yo & 1) << 1) | (xo & 1);
    const int64_t base = ((int64_t)(b * q.g.Hp + (yo >> 1)) * q.g.Wp + (xo >> 1)) * q.g.Co;
#pragma unroll
    for (int p = 0; p < NV; ++p)
      r[p] = (mok && co[p] < q.g.Co) ? route4(q.dp, q.am, base + co[p], j) : f4zero();
  }
};

template <class Cfg>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(typename WgradA<Cfg::NVA>::Params pa,
                                                         typename WgradB<Cfg::NVB>::Params pb, float* slab,
                                                         int tiles_n, int nk, int ks_per_split) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const TileCoord tc = tile_coord(tiles_n);
  const int m0 = tc.mt * Cfg::BM, n0 = tc.nt * Cfg::BN;
  const int split = blockIdx.y;
  WgradA<Cfg::NVA> al; al.init(pa, m0, tid);
  WgradB<Cfg::NVB> bl; bl.init(pb, n0, tid);
  f32x16 acc[Cfg::TM][Cfg::TN];
  acc_zero<Cfg>(acc);
  const int ks0 = split * ks_per_split;
  const int ks1 = min(nk, ks0 + ks_per_split);
  gemm_mainloop<Cfg>(al, bl, acc, ks0, ks1, pa.g.Mtot, smem);
  const int Co = pa.g.Co;
  float* out = slab + (int64_t)split * pa.KI * Co;
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const int col = n0 + acc_col<Cfg>(wn, j, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + acc_row<Cfg>(wm, i, r, lane);
        if (row < pa.KI && col < Co) out[(int64_t)row * Co + col] = acc[i][j][r];
      }
    }
}

// slab[split][(ky,kx,ciP)][co]  ->  dw[co][ci][ky][kx]
__global__ void wgrad_reduce_kernel(const float* slab, float* dw, int splits, int KI, int CiP, int Ci, int Co) {
  const int total = Co * Ci * 9;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int tap = e % 9;
  const int t = e / 9;
  const int ci = t % Ci;
  const int co = t / Ci;
  const int64_t src = (int64_t)(tap * CiP + ci) * Co + co;
  float v = 0.f;
  for (int s = 0; s < splits; ++s) v += slab[(int64_t)s * KI * Co + src];
  dw[e] = v;
}

__global__ void pack_weights_kernel(const float* w, float* wf, float* wd, int Co, int Ci, int CiP) {
  const int total = 9 * CiP * Co;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  // e indexes wf: [(tap*CiP + ci)][co]
  const int co = e % Co;
  const int t = e / Co;
  const int ci = t % CiP;
  const int tap = t / CiP;
  const float v = ci < Ci ? w[((int64_t)co * Ci + ci) * 9 + tap] : 0.f;
  wf[e] = v;
  if (wd) wd[((int64_t)tap * Co + co) * CiP + ci] = v;
}

__global__ void nchw_to_nhwc4_kernel(const float* x, float* y, int C, int64_t HW, int64_t total) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / HW, p = e - b * HW;
    const float* s = x + b * C * HW + p;
    float4 v;
    v.x = s[0];
    v.y = C > 1 ? s[HW] : 0.f;
    v.z = C > 2 ? s[2 * HW] : 0.f;
    v.w = C > 3 ? s[3 * HW] : 0.f;
    reinterpret_cast<float4*>(y)[e] = v;
  }
}

using Cfg128 = TileCfg<128, 128, 2, 2>;
using Cfg128x64 = TileCfg<128, 64, 2, 2>;
using Cfg64 = TileCfg<64, 64, 2, 2>;

template <class K>
static int set_smem(K kern, int bytes, const char* what) {
  return check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes), what);
}

template <class Cfg>
static int launch_fwd(const float* x, const float* wf, const float* bias, float* pooled, uint8_t* amax,
                      const ConvGeom& g, hipStream_t s) {
  const int nWin = g.B * g.Hp * g.Wp, K = 9 * g.CiP;
  typename ConvFwdA<Cfg::NVA>::Params pa{x, g.H, g.W, g.CiP, g.Hp, g.Wp, g.stride, nWin, K};
  typename PlainC<Cfg::NVB>::Params pb{wf, g.Co, g.Co, K};
  const int tiles_m = (4 * nWin + Cfg::BM - 1) / Cfg::BM, tiles_n = (g.Co + Cfg::BN - 1) / Cfg::BN;
  auto kern = conv_fwd_kernel<Cfg>;
  static bool done = false;
  if (!done) { int rc = set_smem(kern, Cfg::SMEM_BYTES, "attr(conv_fwd)"); if (rc) return rc; done = true; }
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), Cfg::SMEM_BYTES, s, pa, pb, bias, pooled, amax,
                     g.Co, tiles_n, (K + BK - 1) / BK);
  return check_hip(hipGetLastError(), "conv_fwd launch");
}

template <class Cfg>
static int launch_dgrad(const float* dp, const uint8_t* am, const float* wd, float* dx, const ConvGeom& g,
                        hipStream_t s) {
  const int rows = g.B * g.H * g.W, K = 9 * g.Co;
  typename ConvDgradA<Cfg::NVA>::Params pa{dp, am, g.H, g.W, g.Hp, g.Wp, g.Co, g.stride, rows, K};
  typename PlainC<Cfg::NVB>::Params pb{wd, g.CiP, g.CiP, K};
  const int tiles_m = (rows + Cfg::BM - 1) / Cfg::BM, tiles_n = (g.CiP + Cfg::BN - 1) / Cfg::BN;
  auto kern = conv_dgrad_kernel<Cfg>;
  static bool done = false;
  if (!done) { int rc = set_smem(kern, Cfg::SMEM_BYTES, "attr(conv_dgrad)"); if (rc) return rc; done = true; }
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), Cfg::SMEM_BYTES, s, pa, pb, dx, g.CiP, tiles_n,
                     (K + BK - 1) / BK);
  return check_hip(hipGetLastError(), "conv_dgrad launch");
}

struct WgradPlan { int big, tiles_m, tiles_n, nk, splits, ks_per_split, Mtot, KI; };
static WgradPlan plan_wgrad(const ConvGeom& g) {
  WgradPlan p;
  p.KI = 9 * g.CiP;
  p.Mtot = g.B * 2 * g.Hp * 2 * g.Wp;
  p.big = (p.KI >= 128 && g.Co >= 128) ? 1 : 0;
  const int bm = p.big ? 128 : 64;
  p.tiles_m = (p.KI + bm - 1) / bm;
  p.tiles_n = (g.Co + bm - 1) / bm;
  p.nk = (p.Mtot + BK - 1) / BK;
  const int tiles = p.tiles_m * p.tiles_n;
  int splits = (512 + tiles - 1) / tiles;
  const int max_splits = p.nk / 8 > 1 ? p.nk / 8 : 1;
  if (splits > max_splits) splits = max_splits;
  p.ks_per_split = (p.nk + splits - 1) / splits;
  p.splits = (p.nk + p.ks_per_split - 1) / p.ks_per_split;
  return p;
}

template <class Cfg>
static int launch_wgrad(const float* x, const float* dp, const uint8_t* am, float* slab, const ConvGeom& g,
                        const WgradPlan& p, hipStream_t s) {
  WgradGeom wg{g.H, g.W, g.CiP, g.Hp, g.Wp, g.Co, g.stride, p.Mtot};
  typename WgradA<Cfg::NVA>::Params pa{x, wg, p.KI};
  typename WgradB<Cfg::NVB>::Params pb{dp, am, wg};
  auto kern = conv_wgrad_kernel<Cfg>;
  static bool done = false;
  if (!done) { int rc = set_smem(kern, Cfg::SMEM_BYTES, "attr(conv_wgrad)"); if (rc) return rc; done = true; }
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, p.splits), dim3(256), Cfg::SMEM_BYTES, s, pa, pb, slab,
                     p.tiles_n, p.nk, p.ks_per_split);
  return check_hip(hipGetLastError(), "conv_wgrad launch");
}

static int check_geom(const char* fn, const ConvGeom& g) {
  VQA_REQUIRE(g.B > 0 && g.H >= 3 && g.W >= 3, "%s: bad image shape B=%d H=%d W=%d", fn, g.B, g.H, g.W);
  VQA_REQUIRE(g.CiP % 4 == 0 && g.Co % 4 == 0 && g.CiP > 0 && g.Co > 0,
              "%s: channel counts must be positive multiples of 4 (CiP=%d Co=%d)", fn, g.CiP, g.Co);
  VQA_REQUIRE(g.stride == 1 || g.stride == 2, "%s: stride %d unsupported (1 or 2)", fn, g.stride);
  VQA_REQUIRE(g.Hp > 0 && g.Wp > 0, "%s: image too small for conv+pool", fn);
  VQA_REQUIRE((int64_t)g.B * g.H * g.W < (1LL << 31) / 4, "%s: too many pixels for 32-bit row indices", fn);
  return VQA_OK;
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

int vqa_conv3x3_relu_pool_fwd(const float* x, const float* wf, const float* bias, float* pooled, uint8_t* argmax,
                              int B, int H, int W, int CiP, int Co, int stride, int tag, vqa_stream_t stream) {
  VQA_REQUIRE(x && wf && bias && pooled && argmax, "vqa_conv3x3_relu_pool_fwd: null pointer");
  const ConvGeom g = make_geom(B, H, W, CiP, Co, stride);
  int rc = check_geom("vqa_conv3x3_relu_pool_fwd", g);
  if (rc) return rc;
  set_launch_tag(tag);
  ProfScope prof(VQA_K_CONV_FWD, (hipStream_t)stream);
  if (Co > 64) return launch_fwd<Cfg128>(x, wf, bias, pooled, argmax, g, (hipStream_t)stream);
  return launch_fwd<Cfg128x64>(x, wf, bias, pooled, argmax, g, (hipStream_t)stream);
}

int vqa_conv3x3_dgrad(const float* dpooled, const uint8_t* argmax, const float* wd, float* dx, int B, int H, int W,
                      int CiP, int Co, int stride, int tag, vqa_stream_t stream) {
  VQA_REQUIRE(dpooled && argmax && wd && dx, "vqa_conv3x3_dgrad: null pointer");
  const ConvGeom g = make_geom(B, H, W, CiP, Co, stride);
  int rc = check_geom("vqa_conv3x3_dgrad", g);
  if (rc) return rc;
  set_launch_tag(tag);
  ProfScope prof(VQA_K_CONV_DGRAD, (hipStream_t)stream);
  if (CiP > 64) return launch_dgrad<Cfg128>(dpooled, argmax, wd, dx, g, (hipStream_t)stream);
  return launch_dgrad<Cfg128x64>(dpooled, argmax, wd, dx, g, (hipStream_t)stream);
}

int64_t vqa_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int CiP, int Co, int stride) {
  const ConvGeom g = make_geom(B, H, W, CiP, Co, stride);
  if (g.Hp <= 0 || g.Wp <= 0) return 0;
  const WgradPlan p = plan_wgrad(g);
  const int64_t slab = (int64_t)p.splits * p.KI * Co * 4;
  const int64_t cs = colsum_ws_bytes((int64_t)B * g.Hp * g.Wp, Co);
  return slab + cs;
}

int vqa_conv3x3_wgrad(const float* x, const float* dpooled, const uint8_t* argmax, float* dw, float* dbias, int B,
                      int H, int W, int CiP, int Ci, int Co, int stride, float* workspace, int64_t workspace_bytes,
                      int tag, vqa_stream_t stream) {
  VQA_REQUIRE(x && dpooled && argmax && dw && dbias && workspace, "vqa_conv3x3_wgrad: null pointer");
  const ConvGeom g = make_geom(B, H, W, CiP, Co, stride);
  int rc = check_geom("vqa_conv3x3_wgrad", g);
  if (rc) return rc;
  VQA_REQUIRE(Ci >= 1 && Ci <= CiP, "vqa_conv3x3_wgrad: Ci=%d CiP=%d", Ci, CiP);
  const WgradPlan p = plan_wgrad(g);
  const int64_t slab_bytes = (int64_t)p.splits * p.KI * Co * 4;
  const int64_t need = vqa_conv3x3_wgrad_workspace_bytes(B, H, W, CiP, Co, stride);
  if (workspace_bytes < need) {
    set_error("vqa_conv3x3_wgrad: workspace %lld < %lld", (long long)workspace_bytes, (long long)need);
    return VQA_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  set_launch_tag(tag);
  {
    ProfScope prof(VQA_K_CONV_WGRAD, s);
    rc = p.big ? launch_wgrad<Cfg128>(x, dpooled, argmax, workspace, g, p, s)
               : launch_wgrad<Cfg64>(x, dpooled, argmax, workspace, g, p, s);
    if (rc) return rc;
    const int total = Co * Ci * 9;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, workspace, dw, p.splits,
                       p.KI, CiP, Ci, Co);
    rc = check_hip(hipGetLastError(), "wgrad_reduce launch");
    if (rc) return rc;
  }
  float* cs_ws = workspace + slab_bytes / 4;
  return colsum_launch(dpooled, Co, argmax, (int64_t)B * g.Hp * g.Wp, Co, dbias, 0, cs_ws,
                       workspace_bytes - slab_bytes, s);
}

}  // extern "C"
