// Convolution blocks with image.kernel_size != 3 (models/model.py:75-82 builds nn.Conv2d(kernel_size=k, stride) -> ReLU ->
// MaxPool2d(2,2) for whatever k the schema admits, utils/config_schema.py:59; config.yaml:58 ships k = 3).
//
// The 3x3 blocks have their own implicit-GEMM / patch kernels (conv.hip, conv0.hip, conv_patch_*.hip).  Every other kernel size
// takes the MATERIALISED form of the same product: the im2col matrix [B*Ho*Wo][k*k*CiP] is written once per batch chunk, the
// contraction is the library's GEMM engine (vqa_gemm: fp32 MFMA, bias in its epilogue), and the four passes here are the
// HBM-bound pieces around it -- all with 16-byte accesses over the channel axis (CiP % 4 == 0, Co % 4 == 0):
//   vqa_convk_im2col     x NHWC            -> cols [rows][k*k*CiP]          K order (ky, kx, ci) = the packed weight's
//   vqa_convk_relu_pool  y [rows][Co]      -> pooled NHWC + arg-max bytes    (same encoding as the 3x3 kernels: 0..3, 4 = dead)
//   vqa_convk_route      dpooled, arg-max  -> dY [rows][Co]                  (pre-pool gradient; rows outside every window: 0)
//   vqa_convk_col2im     dcols             -> dX NHWC                        (gather form: no atomics, deterministic)
// Not a fast path (the im2col matrix costs k*k times the activation's bytes); it exists so that no value of the schema's
// kernel_size is refused.  bound: HBM for the passes, MFMA for the GEMM.
#include "common.hpp"

namespace vqa {

static inline int grid_for(int64_t n, int per_block, int cap = 1 << 20) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

// cols[r][(ky*ks + kx)*CiP + c] = x[b][yo*stride + ky][xo*stride + kx][c],  r = (b*Ho + yo)*Wo + xo
__global__ __launch_bounds__(256) void convk_im2col_kernel(const float4* __restrict__ x, float4* __restrict__ cols, int64_t total4,
                                                           int H, int W, int C4, int ks, int stride, int Ho, int Wo) {
  const int K4 = ks * ks * C4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / K4;
    const int k4 = (int)(e - r * K4);
    const int tap = k4 / C4, c4 = k4 - tap * C4;
    const int ky = tap / ks, kx = tap - ky * ks;
    const int xo = (int)(r % Wo);
    const int64_t t = r / Wo;
    const int yo = (int)(t % Ho);
    const int64_t b = t / Ho;
    cols[e] = x[((b * H + yo * stride + ky) * W + xo * stride + kx) * C4 + c4];
  }
}

// one thread: one pool window x 4 channels.  y is the convolution output INCLUDING the bias, before the ReLU:
// max(relu(.)) == relu(max(.)) and the first strict maximum is what nn.MaxPool2d returns.
__global__ __launch_bounds__(256) void convk_relu_pool_kernel(const float4* __restrict__ y, float4* __restrict__ pooled,
                                                              uint32_t* __restrict__ amax, int64_t total4, int Ho, int Wo,
                                                              int Hp, int Wp, int Co4) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % Co4);
    const int64_t wdw = e / Co4;
    const int px = (int)(wdw % Wp);
    const int64_t t = wdw / Wp;
    const int py = (int)(t % Hp);
    const int64_t b = t / Hp;
    const int64_t r0 = (b * Ho + 2 * py) * Wo + 2 * px;
    const float4 v0 = y[r0 * Co4 + c4], v1 = y[(r0 + 1) * Co4 + c4];
    const float4 v2 = y[(r0 + Wo) * Co4 + c4], v3 = y[(r0 + Wo + 1) * Co4 + c4];
    const float a0[4] = {v0.x, v0.y, v0.z, v0.w}, a1[4] = {v1.x, v1.y, v1.z, v1.w};
    const float a2[4] = {v2.x, v2.y, v2.z, v2.w}, a3[4] = {v3.x, v3.y, v3.z, v3.w};
    float o[4];
    uint32_t am = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float best = a0[i];
      uint32_t a = 0;
      if (a1[i] > best) { best = a1[i]; a = 1; }
      if (a2[i] > best) { best = a2[i]; a = 2; }
      if (a3[i] > best) { best = a3[i]; a = 3; }
      o[i] = best > 0.f ? best : 0.f;
      am |= (best > 0.f ? a : 4u) << (8 * i);
    }
    pooled[e] = make_float4(o[0], o[1], o[2], o[3]);
    amax[e] = am;
  }
}

// dY[(b, yo, xo)][c] = dpooled[(b, yo/2, xo/2)][c] if the window's arg-max byte names this pixel, else 0
__global__ __launch_bounds__(256) void convk_route_kernel(const float4* __restrict__ dpooled, const uint32_t* __restrict__ amax,
                                                          float4* __restrict__ dy, int64_t total4, int Ho, int Wo, int Hp, int Wp,
                                                          int Co4) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % Co4);
    const int64_t r = e / Co4;
    const int xo = (int)(r % Wo);
    const int64_t t = r / Wo;
    const int yo = (int)(t % Ho);
    const int64_t b = t / Ho;
    const int py = yo >> 1, px = xo >> 1;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (py < Hp && px < Wp) {
      const int64_t w = ((b * Hp + py) * Wp + px) * Co4 + c4;
      const uint32_t j = (uint32_t)(((yo & 1) << 1) | (xo & 1));
      const uint32_t am = amax[w];
      const float4 g = dpooled[w];
      o.x = (am & 0xffu) == j ? g.x : 0.f;
      o.y = ((am >> 8) & 0xffu) == j ? g.y : 0.f;
      o.z = ((am >> 16) & 0xffu) == j ? g.z : 0.f;
      o.w = (am >> 24) == j ? g.w : 0.f;
    }
    dy[e] = o;
  }
}

// dX[b][y][x][c] = sum over the taps (ky, kx) whose output pixel ((y-ky)/stride, (x-kx)/stride) exists of
// dcols[that pixel][(ky*ks + kx)*CiP + c]; taps in ascending order (a fixed summation order)
__global__ __launch_bounds__(256) void convk_col2im_kernel(const float4* __restrict__ dcols, float4* __restrict__ dx, int64_t total4,
                                                           int H, int W, int C4, int ks, int stride, int Ho, int Wo) {
  const int64_t K4 = (int64_t)ks * ks * C4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % C4);
    const int64_t p = e / C4;
    const int xx = (int)(p % W);
    const int64_t t = p / W;
    const int yy = (int)(t % H);
    const int64_t b = t / H;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < ks; ++ky) {
      const int ty = yy - ky;
      if (ty < 0 || ty % stride) continue;
      const int yo = ty / stride;
      if (yo >= Ho) continue;
      for (int kx = 0; kx < ks; ++kx) {
        const int tx = xx - kx;
        if (tx < 0 || tx % stride) continue;
        const int xo = tx / stride;
        if (xo >= Wo) continue;
        const float4 g = dcols[((b * Ho + yo) * Wo + xo) * K4 + (int64_t)(ky * ks + kx) * C4 + c4];
        s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
      }
    }
    dx[e] = s;
  }
}

// w [Co][Ci][ks][ks] (nn.Conv2d) <-> wk [Co][(ky*ks + kx)*CiP + ci]; channels ci >= Ci of wk are zero
__global__ __launch_bounds__(256) void convk_pack_kernel(const float* __restrict__ w, float* __restrict__ wk, int64_t total, int Ci,
                                                         int CiP, int ks) {
  const int KK = ks * ks;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(e % CiP);
    const int64_t t = e / CiP;
    const int tap = (int)(t % KK);
    const int64_t co = t / KK;
    wk[e] = ci < Ci ? w[(co * Ci + ci) * KK + tap] : 0.f;
  }
}
__global__ __launch_bounds__(256) void convk_unpack_kernel(const float* __restrict__ dwk, float* __restrict__ dw, int64_t total, int Ci,
                                                           int CiP, int ks) {
  const int KK = ks * ks;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int tap = (int)(e % KK);
    const int64_t t = e / KK;
    const int ci = (int)(t % Ci);
    const int64_t co = t / Ci;
    dw[e] = dwk[(co * KK + tap) * CiP + ci];
  }
}

static int convk_geom(const char* fn, int B, int H, int W, int C, int ks, int stride, int* Ho, int* Wo) {
  VQA_REQUIRE(B > 0 && C > 0 && C % 4 == 0, "%s: B=%d, channels=%d (positive, channels a multiple of 4)", fn, B, C);
  VQA_REQUIRE(ks >= 1 && ks <= 15 && (stride == 1 || stride == 2), "%s: kernel_size %d (1..15), stride %d (1 or 2)", fn, ks, stride);
  VQA_REQUIRE(H >= ks && W >= ks, "%s: image %dx%d smaller than the %dx%d kernel", fn, H, W, ks, ks);
  *Ho = (H - ks) / stride + 1;
  *Wo = (W - ks) / stride + 1;
  return VQA_OK;
}
#define ALIGNED16(p) (((uintptr_t)(p) % 16) == 0)

}  // namespace vqa

using namespace vqa;
#define STREAM ((hipStream_t)stream)

extern "C" {

int vqa_convk_pack_weights(const float* w, float* wk, int Co, int Ci, int CiP, int ks, vqa_stream_t stream) {
  VQA_REQUIRE(w && wk && Co > 0 && Ci > 0 && CiP >= Ci && CiP % 4 == 0 && ks >= 1 && ks <= 15,
              "vqa_convk_pack_weights: bad args (Co=%d Ci=%d CiP=%d ks=%d)", Co, Ci, CiP, ks);
  const int64_t total = (int64_t)Co * ks * ks * CiP;
  hipLaunchKernelGGL(convk_pack_kernel, dim3(grid_for(total, 256)), dim3(256), 0, STREAM, w, wk, total, Ci, CiP, ks);
  return check_hip(hipGetLastError(), "convk_pack launch");
}

int vqa_convk_unpack_wgrad(const float* dwk, float* dw, int Co, int Ci, int CiP, int ks, vqa_stream_t stream) {
  VQA_REQUIRE(dwk && dw && Co > 0 && Ci > 0 && CiP >= Ci && CiP % 4 == 0 && ks >= 1 && ks <= 15,
              "vqa_convk_unpack_wgrad: bad args (Co=%d Ci=%d CiP=%d ks=%d)", Co, Ci, CiP, ks);
  const int64_t total = (int64_t)Co * Ci * ks * ks;
  hipLaunchKernelGGL(convk_unpack_kernel, dim3(grid_for(total, 256)), dim3(256), 0, STREAM, dwk, dw, total, Ci, CiP, ks);
  return check_hip(hipGetLastError(), "convk_unpack launch");
}

int vqa_convk_im2col(const float* x, float* cols, int B, int H, int W, int CiP, int ks, int stride, vqa_stream_t stream) {
  int Ho, Wo;
  if (int rc = convk_geom("vqa_convk_im2col", B, H, W, CiP, ks, stride, &Ho, &Wo)) return rc;
  VQA_REQUIRE(x && cols && ALIGNED16(x) && ALIGNED16(cols), "vqa_convk_im2col: null or unaligned pointer");
  const int64_t total4 = (int64_t)B * Ho * Wo * ks * ks * (CiP / 4);
  hipLaunchKernelGGL(convk_im2col_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, STREAM, reinterpret_cast<const float4*>(x),
                     reinterpret_cast<float4*>(cols), total4, H, W, CiP / 4, ks, stride, Ho, Wo);
  return check_hip(hipGetLastError(), "convk_im2col launch");
}

int vqa_convk_relu_pool(const float* y, float* pooled, uint8_t* amax, int B, int Ho, int Wo, int Co, vqa_stream_t stream) {
  VQA_REQUIRE(y && pooled && amax && ALIGNED16(y) && ALIGNED16(pooled) && ((uintptr_t)amax % 4) == 0,
              "vqa_convk_relu_pool: null or unaligned pointer");
  VQA_REQUIRE(B > 0 && Ho >= 2 && Wo >= 2 && Co > 0 && Co % 4 == 0, "vqa_convk_relu_pool: bad shape B=%d Ho=%d Wo=%d Co=%d", B, Ho,
              Wo, Co);
  const int Hp = Ho / 2, Wp = Wo / 2;
  const int64_t total4 = (int64_t)B * Hp * Wp * (Co / 4);
  hipLaunchKernelGGL(convk_relu_pool_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, STREAM, reinterpret_cast<const float4*>(y),
                     reinterpret_cast<float4*>(pooled), reinterpret_cast<uint32_t*>(amax), total4, Ho, Wo, Hp, Wp, Co / 4);
  return check_hip(hipGetLastError(), "convk_relu_pool launch");
}

int vqa_convk_route(const float* dpooled, const uint8_t* amax, float* dy, int B, int Ho, int Wo, int Co, vqa_stream_t stream) {
  VQA_REQUIRE(dpooled && amax && dy && ALIGNED16(dpooled) && ALIGNED16(dy) && ((uintptr_t)amax % 4) == 0,
              "vqa_convk_route: null or unaligned pointer");
  VQA_REQUIRE(B > 0 && Ho >= 2 && Wo >= 2 && Co > 0 && Co % 4 == 0, "vqa_convk_route: bad shape B=%d Ho=%d Wo=%d Co=%d", B, Ho, Wo, Co);
  const int64_t total4 = (int64_t)B * Ho * Wo * (Co / 4);
  hipLaunchKernelGGL(convk_route_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, STREAM, reinterpret_cast<const float4*>(dpooled),
                     reinterpret_cast<const uint32_t*>(amax), reinterpret_cast<float4*>(dy), total4, Ho, Wo, Ho / 2, Wo / 2, Co / 4);
  return check_hip(hipGetLastError(), "convk_route launch");
}

int vqa_convk_col2im(const float* dcols, float* dx, int B, int H, int W, int CiP, int ks, int stride, vqa_stream_t stream) {
  int Ho, Wo;
  if (int rc = convk_geom("vqa_convk_col2im", B, H, W, CiP, ks, stride, &Ho, &Wo)) return rc;
  VQA_REQUIRE(dcols && dx && ALIGNED16(dcols) && ALIGNED16(dx), "vqa_convk_col2im: null or unaligned pointer");
  const int64_t total4 = (int64_t)B * H * W * (CiP / 4);
  hipLaunchKernelGGL(convk_col2im_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, STREAM, reinterpret_cast<const float4*>(dcols),
                     reinterpret_cast<float4*>(dx), total4, H, W, CiP / 4, ks, stride, Ho, Wo);
  return check_hip(hipGetLastError(), "convk_col2im launch");
}

}  // extern "C"
