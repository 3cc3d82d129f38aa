// Convolution blocks in fp32 on the bf16 matrix cores (x3_core.hpp: exact three-way bf16 split, six partial products,
// fp32 accumulate).  Same tensors, layouts, loaders and epilogues as conv.hip -- only the K loop differs -- so the
// entry points take exactly the arguments of their conv.hip counterparts.  Reference: models/model.py:72-84.
#include "x3_core.hpp"

namespace vqa {

#include "conv_device.inc"
#include "conv_host.inc"

#ifndef VQA_X3_PF
#define VQA_X3_PF 2
#endif
// one workgroup per CU (240 bytes of LDS per tile row and stage): 4 MFMA waves + 4 loader waves, 256 VGPRs each
using CfgX = TileCfg<192, 128, 2, 2, 4, VQA_X3_PF>;      // MFMA waves of 96 x 64
using CfgXn = TileCfg<256, 64, 4, 1, 4, VQA_X3_PF>;      // 64 output columns (conv1 dgrad): MFMA waves of 64 x 64
// K-steps of loads in flight per kernel family (same box, packed operands, conv1 / conv2 ms): dgrad 1: 2.75 / 1.83, 2: 2.82 /
// 1.88, 3 spills (5.7 / 2.4); wgrad 1: 2.05 / 1.85, 2: 1.94 / 1.72, 3: 1.89 / 1.67; forward the same at 1, 2 and 3
using CfgXd = TileCfg<192, 128, 2, 2, 4, 1>;
#ifndef VQA_X3_ND16
#define VQA_X3_ND16 1
#endif
#if VQA_X3_ND16
// 64 output columns (conv1 dgrad): every routed A element feeds only 64 columns, so the loaders are the long pole (MfmaUtil
// 38 % with 4 + 4 waves).  16 waves: 8 MFMA waves of 64 x 32 (104 VGPRs) + 8 loader waves, four waves per SIMD.
using CfgXnd = TileCfg<256, 64, 4, 2, 8, 1>;
#else
using CfgXnd = TileCfg<256, 64, 4, 1, 4, 1>;
#endif
using CfgXw = TileCfg<192, 128, 2, 2, 4, 3>;

// ------------------------------------------------------------------ activations split ahead of time ("x3-packed")
// An activation tensor [pixels][C] (C % 4 == 0) can be handed over already split, so that the split is done once per
// tensor (vqa_x3_pack) instead of by every workgroup that reads it (9 taps x column tiles in forward, row tiles in
// wgrad).  Layout: every group of four consecutive channels is 24 bytes, hi[4] mid[4] lo[4] bf16 -- element e lives at
// byte (e / 4) * 24, a pixel's channels stay contiguous (6 bytes per element), and a loader thread's 4-element chunk is
// one 16-byte + one 8-byte load and three ds_write_b64 without any VALU work.  The loaders below are ConvFwdA<UT> and
// WgradA<uniform> with every byte offset scaled by 6 / 4.
template <int NV, int LT>
struct ConvFwdAx {
  struct Params { const void* x; int H, W, CiP, Hp, Wp, stride, nWin, K; };
  struct Raw { float4 a[NV]; uint2 b[NV]; };
  static constexpr bool kTypeR = true;
  static constexpr bool kPreSplit = true;
  const char* x;
  uint32_t voff[NV];
  int W, CiP;
  TapCursor cur;
  __device__ __forceinline__ void init(const Params& q, int row0, int tid, int ks0) {
    x = static_cast<const char*>(q.x); W = q.W; CiP = q.CiP;
    const int c4 = 4 * StageMap<LT>::r_chunk(tid);
    cur.init(ks0 * BK, q.CiP);
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int m = row0 + StageMap<LT>::r_row(tid, p);
      const int wl = m >> 2, j = m & 3;
      const int px = wl % q.Wp;
      const int t = wl / q.Wp;
      const int py = t % q.Hp;
      const int b = t / q.Hp;
      const int y = (2 * py + (j >> 1)) * q.stride, xx = (2 * px + (j & 1)) * q.stride;
      const uint32_t e = (uint32_t)((b * q.H + y) * q.W + xx) * (uint32_t)q.CiP + (uint32_t)c4;
      voff[p] = wl < q.nWin ? e * 6u : BUF_OOB;
    }
  }
  __device__ __forceinline__ void issue(int /*ks*/, Raw& r) {
    const bool kok = cur.tap < 9;
    const int tap = kok ? cur.tap : 0;
    const int ky = tap / 3, kx = tap - 3 * ky;
    const int koff = (ky * W + kx) * CiP + cur.ch;
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(x + (int64_t)koff * 6, kok ? BUF_OOB : 0u);
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      r.a[p] = buf_load16(rs, voff[p]);
      r.b[p] = buf_load8(rs, voff[p], 16);
    }
    cur.advance();
  }
  __device__ __forceinline__ void planes(const Raw& r, int p, uint2& hi, uint2& mid, uint2& lo) const {
    hi = make_uint2(__float_as_uint(r.a[p].x), __float_as_uint(r.a[p].y));
    mid = make_uint2(__float_as_uint(r.a[p].z), __float_as_uint(r.a[p].w));
    lo = r.b[p];
  }
};

template <int NV, int LT>
struct WgradAx {   // A(i = (ky,kx,ci), m) = x[b][yo*s+ky][xo*s+kx][ci], x3-packed
  struct Params { const void* x; WgradGeom g; int KI; };
  struct Raw { float4 a[NV]; uint2 b[NV]; };
  static constexpr bool kTypeR = false;
  static constexpr bool kPreSplit = true;
  Params q;
  uint32_t su[NV];
  uint32_t lp0, lp1;
  uint32_t pix, row, img, fix_img;
  int kr;
  RowCursor rc;
  __device__ __forceinline__ void init(const Params& q_, int i0, int tid, int ks0) {
    q = q_;
    kr = StageMap<LT>::c_krow(tid);
    const int hw = __builtin_amdgcn_readfirstlane(tid >> 8);
    rc.init(ks0 * BK, q.g.Hp, q.g.Wp);
    pix = (uint32_t)(q.g.stride * q.g.CiP) * 6u; row = (uint32_t)q.g.W * pix;
    img = (uint32_t)(q.g.H * q.g.W * q.g.CiP) * 6u;
    fix_img = img - (uint32_t)rc.Ho2 * row;
    lp0 = (uint32_t)kr * pix + 24u * (uint32_t)(tid & 7);
    lp1 = lp0 + row - (uint32_t)rc.Wo2 * pix;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int i = i0 + BK * (hw + (LT / 256) * p);
      const int ii = i < q.KI ? i : 0;
      const int tap = ii / q.g.CiP;
      const int ci = ii - tap * q.g.CiP;
      const int ky = tap / 3, kx = tap - 3 * ky;
      su[p] = (uint32_t)((ky * q.g.W + kx) * q.g.CiP + ci) * 6u;
    }
  }
  __device__ __forceinline__ void issue(int ks, Raw& r) {
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(q.x);
    const int n1 = rc.Wo2 - rc.xs;
    const uint32_t base = (uint32_t)rc.b * img + (uint32_t)rc.yo * row + (uint32_t)rc.xs * pix;
    uint32_t l1 = lp1;
    if (rc.yo + 1 == rc.Ho2 && n1 < BK) l1 += fix_img;
    const uint32_t off = kr < n1 ? lp0 : l1;
    if ((ks + 1) * BK <= q.g.Mtot) {
#pragma unroll
      for (int p = 0; p < NV; ++p) {
        r.a[p] = buf_load16(rs, off, base + su[p]);
        r.b[p] = buf_load8(rs, off, base + su[p] + 16u);
      }
    } else {
      const uint32_t o2 = rc.m0 + kr < q.g.Mtot ? off + base : BUF_OOB;
#pragma unroll
      for (int p = 0; p < NV; ++p) {
        r.a[p] = buf_load16(rs, o2, su[p]);
        r.b[p] = buf_load8(rs, o2, su[p] + 16u);
      }
    }
    rc.advance();
  }
  __device__ __forceinline__ void planes(const Raw& r, int p, uint2& hi, uint2& mid, uint2& lo) const {
    hi = make_uint2(__float_as_uint(r.a[p].x), __float_as_uint(r.a[p].y));
    mid = make_uint2(__float_as_uint(r.a[p].z), __float_as_uint(r.a[p].w));
    lo = r.b[p];
  }
};

// ------------------------------------------------------------------ routed operands from an x3-packed pooled gradient
// dgrad's A and wgrad's B rebuild dY from the pooled gradient and the arg-max bytes (conv_device.inc route4).  On the
// packed form the routing is a mask on 16-bit halves, shared by the three planes: 8 VALU for the masks of four elements
// + 6 ANDs, against 8 (route4) + 14 (split4) on an fp32 gradient.
// id = the four arg-max bytes of the chunk, jrep = the pixel's position in its window in every byte.
__device__ __forceinline__ void route_masks(uint32_t id, uint32_t jrep, uint32_t& m01, uint32_t& m23) {
  const uint32_t t = id ^ jrep;                               // bytes 0..7; zero where this pixel is the arg-max
  const uint32_t g = (0x08080808u - t) & 0x08080808u;         // 0x08 per matching byte (no borrows: every byte of t < 8)
  const uint32_t b = (g << 5) - (g >> 3);                     // 0xff per matching byte
  m01 = __builtin_amdgcn_perm(b, b, 0x01010000u);             // bytes (b1 b1 b0 b0): element 0 is the low half
  m23 = __builtin_amdgcn_perm(b, b, 0x03030202u);
}
__device__ __forceinline__ void route_planes(const float4 a, const uint2 bb, uint32_t id, uint32_t jrep, uint2& hi,
                                             uint2& mid, uint2& lo) {
  uint32_t m01, m23;
  route_masks(id, jrep, m01, m23);
  hi = make_uint2(__float_as_uint(a.x) & m01, __float_as_uint(a.y) & m23);
  mid = make_uint2(__float_as_uint(a.z) & m01, __float_as_uint(a.w) & m23);
  lo = make_uint2(bb.x & m01, bb.y & m23);
}

// ConvDgradA<UT> over an x3-packed pooled gradient (byte offsets x 6 / 4; the arg-max offsets stay in elements)
template <int NV, int LT>
struct ConvDgradAx {
  struct Params { const void* dp; const uint8_t* am; int H, W, Hp, Wp, Co, stride, rows, K; };
  struct Raw { float4 a[NV]; uint2 b[NV]; uint32_t id[NV]; uint32_t j[NV]; };
  static constexpr bool kTypeR = true;
  static constexpr bool kPreSplit = true;
  Params q;
  int b[NV], y[NV], x[NV];
  uint32_t vd[NV], va[NV];
  uint32_t jc[NV];             // the row's position in its source window for the current tap, in every byte
  int curtap, c4;
  TapCursor cur;
  __device__ __forceinline__ void init(const Params& q_, int row0, int tid, int ks0) {
    q = q_;
    c4 = 4 * StageMap<LT>::r_chunk(tid);
    cur.init(ks0 * BK, q.Co);
    curtap = -1;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int m = row0 + StageMap<LT>::r_row(tid, p);
      const int mm = m < q.rows ? m : 0;
      x[p] = m < q.rows ? mm % q.W : -1;
      const int t = mm / q.W;
      y[p] = t % q.H;
      b[p] = t / q.H;
      vd[p] = va[p] = BUF_OOB;
      jc[p] = 0;
    }
  }
  __device__ __forceinline__ void retap(int tap, int c) {
    const bool kok = tap < 9;
    const int tp = kok ? tap : 0;
    const int ky = tp / 3, kx = tp - 3 * ky;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      int yy = y[p] - ky, xx = x[p] - kx;
      bool v = kok && yy >= 0 && xx >= 0;
      if (q.stride == 2) { v = v && !(yy & 1) && !(xx & 1); yy >>= 1; xx >>= 1; }
      v = v && yy < 2 * q.Hp && xx < 2 * q.Wp;
      const uint32_t e = (uint32_t)((b[p] * q.Hp + (yy >> 1)) * q.Wp + (xx >> 1)) * (uint32_t)q.Co + (uint32_t)c;
      vd[p] = v ? e * 6u : BUF_OOB;
      va[p] = v ? e : BUF_OOB;
      jc[p] = (uint32_t)(((yy & 1) << 1) | (xx & 1)) * 0x01010101u;
    }
  }
  __device__ __forceinline__ void issue(int /*ks*/, Raw& r) {
    if (cur.tap != curtap) { retap(cur.tap, c4); curtap = cur.tap; }
    const __amdgpu_buffer_rsrc_t rd = buf_rsrc(static_cast<const char*>(q.dp) + (int64_t)cur.ch * 6), ra = buf_rsrc(q.am + cur.ch);
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      r.a[p] = buf_load16(rd, vd[p]);
      r.b[p] = buf_load8(rd, vd[p], 16);
      r.id[p] = AM_LOAD(ra, va[p]);
      r.j[p] = jc[p];
    }
    cur.advance();
  }
  __device__ __forceinline__ void planes(const Raw& r, int p, uint2& hi, uint2& mid, uint2& lo) const {
    route_planes(r.a[p], r.b[p], r.id[p], r.j[p], hi, mid, lo);
  }
};

// WgradB<uniform> over an x3-packed pooled gradient
template <int NV, int LT>
struct WgradBx {
  struct Params { const void* dp; const uint8_t* am; WgradGeom g; };
  struct Raw { float4 a[NV]; uint2 b[NV]; uint32_t id[NV]; uint32_t j; };
  static constexpr bool kTypeR = false;
  static constexpr bool kPreSplit = true;
  Params q;
  uint32_t le0, leE;        // lane constants in ELEMENTS: (kr >> 1) * Co + 4 * (lane & 7); leE = le0 - Wp * Co
  uint32_t j0;
  int kr, n0, hw;
  RowCursor rc;
  __device__ __forceinline__ void init(const Params& q_, int n0_, int tid, int ks0) {
    q = q_; n0 = n0_;
    kr = StageMap<LT>::c_krow(tid);
    hw = __builtin_amdgcn_readfirstlane(tid >> 8);
    rc.init(ks0 * BK, q.g.Hp, q.g.Wp);
    le0 = (uint32_t)(kr >> 1) * (uint32_t)q.g.Co + 4u * (uint32_t)(tid & 7);
    leE = le0 - (uint32_t)q.g.Wp * (uint32_t)q.g.Co;
    j0 = (uint32_t)(kr & 1);
  }
  __device__ __forceinline__ void issue(int ks, Raw& r) {
    // the packed gradient is < 4 GiB, so its element offsets stay below BUF_OOB / 6: the arg-max resource ends there
    const __amdgpu_buffer_rsrc_t rd = buf_rsrc(static_cast<const char*>(q.dp) + (int64_t)n0 * 6),
                                 ra = buf_rsrc(q.am + n0, BUF_OOB / 6u);
    const int n1 = rc.Wo2 - rc.xs;
    const bool odd = rc.yo & 1;
    const int R = rc.b * rc.Ho2 + rc.yo;
    const uint32_t base = (uint32_t)((R >> 1) * q.g.Wp + (rc.xs >> 1)) * (uint32_t)q.g.Co;
    const bool second = kr >= n1;
    uint32_t el = (second && !odd ? leE : le0) + base;
    r.j = (j0 + (second != odd ? 2u : 0u)) * 0x01010101u;
    const bool dead = (ks + 1) * BK > q.g.Mtot && !(rc.m0 + kr < q.g.Mtot);   // last K-step / past the end
    const uint32_t off = dead ? BUF_OOB : el * 6u;
    const uint32_t offa = dead ? BUF_OOB : el;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int grp = hw + (LT / 256) * p;
      const uint32_t pp = n0 + BK * grp < q.g.Co ? grp : 0;
      r.a[p] = buf_load16(rd, off, 192u * pp);
      r.b[p] = buf_load8(rd, off, 192u * pp + 16u);
      r.id[p] = AM_LOAD(ra, offa, 32u * pp);
    }
    rc.advance();
  }
  __device__ __forceinline__ void planes(const Raw& r, int p, uint2& hi, uint2& mid, uint2& lo) const {
    route_planes(r.a[p], r.b[p], r.id[p], r.j, hi, mid, lo);
  }
};

// conv_pool_epilogue with the pooled activation written in the x3-packed form (the next block's input): bit-identical
// to vqa_x3_pack of the fp32 output.
template <class Cfg>
__device__ __forceinline__ void conv_pool_epilogue_x3p(f32x16 (&acc)[Cfg::TM][Cfg::TN], const float* __restrict__ bias,
                                                       void* pooled_, uint8_t* amax, int nWin, int Co, int m0, int n0,
                                                       int wm, int wn, int lane) {
  char* pooled = static_cast<char*>(pooled_);
  const int h = lane >> 5, l31 = lane & 31;
  const bool interior = (m0 + Cfg::BM) / 4 <= nWin && n0 + Cfg::BN <= Co;   // uniform
  const uint32_t vl = (uint32_t)(h * Co + l31);                             // elements (arg-max bytes)
  const uint32_t vx = 6u * (uint32_t)(h * Co) + x3p_lane(l31);              // bytes (packed activation)
  auto body = [&](auto inner) {
    constexpr bool INNER = decltype(inner)::value;
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const int colt = n0 + wn * Cfg::WN + 32 * j;
      const int col = colt + l31;
      const float bv = INNER || col < Co ? bias[INNER || col < Co ? col : 0] : 0.f;
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) {
        const int wt = (m0 + wm * Cfg::WM + 32 * i) >> 2;                   // tile's first window (uniform)
        const int64_t o0 = (int64_t)wt * Co + colt;
        const __amdgpu_buffer_rsrc_t rp = buf_rsrc(pooled + 6 * o0), ra = buf_rsrc(amax + o0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float best = acc[i][j][4 * g];
          int a = 0;
          if (acc[i][j][4 * g + 1] > best) { best = acc[i][j][4 * g + 1]; a = 1; }
          if (acc[i][j][4 * g + 2] > best) { best = acc[i][j][4 * g + 2]; a = 2; }
          if (acc[i][j][4 * g + 3] > best) { best = acc[i][j][4 * g + 3]; a = 3; }
          best += bv;
          const bool ok = INNER || (wt + 2 * g + h < nWin && col < Co);
          const uint32_t so = (uint32_t)(2 * g * Co);
          uint16_t sh, sm, sl;
          split1(best > 0.f ? best : 0.f, sh, sm, sl);
          buf_store2(rp, sh, ok ? vx : BUF_OOB, 6u * so);
          buf_store2(rp, sm, ok ? vx : BUF_OOB, 6u * so + 8u);
          buf_store2(rp, sl, ok ? vx : BUF_OOB, 6u * so + 16u);
          buf_store1(ra, best > 0.f ? (uint8_t)a : (uint8_t)4, ok ? vl : BUF_OOB, so);
        }
      }
    }
  };
  if (interior) body(std::true_type{}); else body(std::false_type{});
}

template <class Cfg, class AL, bool OP>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::THREADS / 256) void conv_fwd_x3_kernel(typename AL::Params pa,
                                                                      typename PlainCx<Cfg::NVB, Cfg::LT>::Params pb,
                                                                      const float* __restrict__ bias, void* pooled,
                                                                      uint8_t* amax, int Co, int tiles_m, int tiles_n, int nk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const TileCoord tc = tile_coord(tiles_m, tiles_n);
  const int m0 = tc.mt * Cfg::BM, n0 = tc.nt * Cfg::BN;
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
  if constexpr (OP) conv_pool_epilogue_x3p<Cfg>(acc, bias, pooled, amax, pa.nWin, Co, m0, n0, wm, wn, lane);
  else conv_pool_epilogue<Cfg>(acc, bias, pooled, amax, pa.nWin, Co, m0, n0, wm, wn, lane);
}

// persistent variants: one workgroup per CU walks the tiles, the loaders run ahead into the next tile during the epilogue
template <class Cfg, class AL, bool OP>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::THREADS / 256) void conv_fwd_x3_persistent_kernel(typename AL::Params pa,
                                                                                 typename PlainCx<Cfg::NVB, Cfg::LT>::Params pb,
                                                                                 const float* __restrict__ bias, void* pooled,
                                                                                 uint8_t* amax, int Co, int tiles_m, int tiles_n,
                                                                                 int nk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  using BL = PlainCx<Cfg::NVB, Cfg::LT>;
  gemm_persistent_x<Cfg, AL, BL>(
      xcd_swizzle(blockIdx.x, gridDim.x), gridDim.x, tiles_m * tiles_n, nk, smem,
      [&](int t, AL& al, BL& bl) {
        const int mt = t / tiles_n, nt = t - mt * tiles_n;
        al.init(pa, mt * Cfg::BM, loader_tid<Cfg>(), 0);
        bl.init(pb, nt * Cfg::BN, loader_tid<Cfg>(), 0);
      },
      [&](int t, f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
        const int mt = t / tiles_n, nt = t - mt * tiles_n;
        if constexpr (OP) conv_pool_epilogue_x3p<Cfg>(acc, bias, pooled, amax, pa.nWin, Co, mt * Cfg::BM, nt * Cfg::BN, wm, wn, lane);
        else conv_pool_epilogue<Cfg>(acc, bias, pooled, amax, pa.nWin, Co, mt * Cfg::BM, nt * Cfg::BN, wm, wn, lane);
      });
}

template <class Cfg, class AL>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::THREADS / 256) void conv_dgrad_x3_persistent_kernel(typename AL::Params pa,
                                                                                   typename PlainCx<Cfg::NVB, Cfg::LT>::Params pb,
                                                                                   float* dx, int CiP, int tiles_m, int tiles_n,
                                                                                   int nk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  using BL = PlainCx<Cfg::NVB, Cfg::LT>;
  gemm_persistent_x<Cfg, AL, BL>(
      xcd_swizzle(blockIdx.x, gridDim.x), gridDim.x, tiles_m * tiles_n, nk, smem,
      [&](int t, AL& al, BL& bl) {
        const int mt = t / tiles_n, nt = t - mt * tiles_n;
        al.init(pa, mt * Cfg::BM, loader_tid<Cfg>(), 0);
        bl.init(pb, nt * Cfg::BN, loader_tid<Cfg>(), 0);
      },
      [&](int t, f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
        const int mt = t / tiles_n, nt = t - mt * tiles_n;
        store_acc_tiles<Cfg>(acc, dx, CiP, pa.rows, CiP, mt * Cfg::BM, nt * Cfg::BN, wm, wn, lane);
      });
}

template <class Cfg, class AL>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::THREADS / 256) void conv_dgrad_x3_kernel(typename AL::Params pa,
                                                                        typename PlainCx<Cfg::NVB, Cfg::LT>::Params pb,
                                                                        float* dx, int CiP, int tiles_m, int tiles_n, int nk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const TileCoord tc = tile_coord(tiles_m, tiles_n);
  const int m0 = tc.mt * Cfg::BM, n0 = tc.nt * Cfg::BN;
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

template <class Cfg, class AL, class BL>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::THREADS / 256) void conv_wgrad_x3_kernel(typename AL::Params pa, typename BL::Params pb,
                                                                        float* slab, int tiles_m, int tiles_n, int nk,
                                                                        int ks_per_split) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const TileCoord tc = tile_coord(tiles_m, tiles_n);
  const int m0 = tc.mt * Cfg::BM, n0 = tc.nt * Cfg::BN;
  const int ks0 = tc.split * ks_per_split;
  const int ks1 = min(nk, ks0 + ks_per_split);
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

template <class Cfg, class AL, bool OP>
static int launch_fwd_x3(const void* x, const void* wf, const float* bias, void* pooled, uint8_t* amax,
                         const ConvGeom& g, hipStream_t s) {
  using SL = SmemLayoutX<Cfg, true, false>;
  const int nWin = g.B * g.Hp * g.Wp, K = 9 * g.CiP;
  typename AL::Params pa{static_cast<decltype(AL::Params::x)>(x), g.H, g.W, g.CiP, g.Hp, g.Wp, g.stride, nWin, K};
  typename PlainCx<Cfg::NVB, Cfg::LT>::Params pb{wf, g.Co, g.Co, K, (int64_t)K * g.Co};
  const int tiles_m = (4 * nWin + Cfg::BM - 1) / Cfg::BM, tiles_n = (g.Co + Cfg::BN - 1) / Cfg::BN;
  // one workgroup per CU: nothing else covers a tile's prologue and epilogue, so the tiles are walked persistently
  // (VQA_PERSISTENT=0 selects one workgroup per tile)
  const int tiles = tiles_m * tiles_n;
  if (knobs().persistent != 0 && tiles > 256) {
    auto pk = conv_fwd_x3_persistent_kernel<Cfg, AL, OP>;
    { int rc = set_smem(pk, SL::BYTES, "attr(conv_fwd_x3_p)"); if (rc) return rc; }
    hipLaunchKernelGGL(pk, dim3(256), dim3(Cfg::THREADS), SL::BYTES, s, pa, pb, bias, pooled, amax, g.Co, tiles_m, tiles_n,
                       K / BK);
    return check_hip(hipGetLastError(), "conv_fwd_x3_persistent launch");
  }
  auto kern = conv_fwd_x3_kernel<Cfg, AL, OP>;
  { int rc = set_smem(kern, SL::BYTES, "attr(conv_fwd_x3)"); if (rc) return rc; }
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(Cfg::THREADS), SL::BYTES, s, pa, pb, bias, pooled, amax, g.Co,
                     tiles_m, tiles_n, K / BK);
  return check_hip(hipGetLastError(), "conv_fwd_x3 launch");
}

template <class Cfg, class AL>
static int launch_dgrad_x3(const void* dp, const uint8_t* am, const void* wd, float* dx, const ConvGeom& g,
                           hipStream_t s) {
  using SL = SmemLayoutX<Cfg, true, false>;
  const int rows = g.B * g.H * g.W, K = 9 * g.Co;
  typename AL::Params pa{static_cast<decltype(AL::Params::dp)>(dp), am, g.H, g.W, g.Hp, g.Wp, g.Co, g.stride, rows, K};
  typename PlainCx<Cfg::NVB, Cfg::LT>::Params pb{wd, g.CiP, g.CiP, K, (int64_t)K * g.CiP};
  const int tiles_m = (rows + Cfg::BM - 1) / Cfg::BM, tiles_n = (g.CiP + Cfg::BN - 1) / Cfg::BN;
  const int tiles = tiles_m * tiles_n;
  if (knobs().persistent != 0 && tiles > 256) {      // see launch_fwd_x3
    auto pk = conv_dgrad_x3_persistent_kernel<Cfg, AL>;
    { int rc = set_smem(pk, SL::BYTES, "attr(conv_dgrad_x3_p)"); if (rc) return rc; }
    hipLaunchKernelGGL(pk, dim3(256), dim3(Cfg::THREADS), SL::BYTES, s, pa, pb, dx, g.CiP, tiles_m, tiles_n, K / BK);
    return check_hip(hipGetLastError(), "conv_dgrad_x3_persistent launch");
  }
  auto kern = conv_dgrad_x3_kernel<Cfg, AL>;
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

// bias gradient = sum over the windows that are not dead (arg-max != 4) of the fp32 pooled gradient; deterministic.  A
// thread owns 4 consecutive channels (one 16-byte load of dP + 4 arg-max bytes per window), the Co/4 threads of a
// window sit side by side, 256 / (Co/4) window lanes per block combine through LDS in a fixed order; the blocks write
// part rows that wgrad_bias_reduce_kernel adds up.  Co % 4 == 0, Co <= 1024.  (The split kernels' B fragments are bf16
// planes: summing them in the MFMA waves as conv.hip does would cost more than this pass over 0.1-0.4 GB.)
// PACK: the same pass also writes the gradient in the x3-packed form (vqa_x3_pack_pooled_grad: one read of dP serves the
// bias gradient and the split).
constexpr int kBiasParts = 512;
template <bool PACK>
__global__ __launch_bounds__(256) void conv_bias_grad_kernel(const float* dp, const uint8_t* am, float* part,
                                                             int64_t windows, int Co, int64_t per, uint2* packed) {
  extern __shared__ __attribute__((aligned(16))) float red[];      // [window lanes][Co]
  const int cpr = Co / 4, nwl = 256 / cpr;
  const int c = threadIdx.x % cpr, wl = threadIdx.x / cpr;
  const int64_t w0 = (int64_t)blockIdx.x * per;
  const int64_t w1 = w0 + per < windows ? w0 + per : windows;
  float4 acc = f4zero();
  const SplitConsts kc = split_consts();
  if (wl < nwl) {
    for (int64_t w = w0 + wl; w < w1; w += nwl) {
      const float4 d = *reinterpret_cast<const float4*>(dp + w * Co + 4 * c);
      const uint32_t a = *reinterpret_cast<const uint32_t*>(am + w * Co + 4 * c);
      if (PACK) {
        uint2 h, m, l;
        split4(d, h, m, l, kc);
        uint2* o = packed + 3 * (w * cpr + c);
        o[0] = h; o[1] = m; o[2] = l;
      }
      acc.x += (a & 0xffu) != 4u ? d.x : 0.f;
      acc.y += ((a >> 8) & 0xffu) != 4u ? d.y : 0.f;
      acc.z += ((a >> 16) & 0xffu) != 4u ? d.z : 0.f;
      acc.w += (a >> 24) != 4u ? d.w : 0.f;
    }
    *reinterpret_cast<float4*>(red + wl * Co + 4 * c) = acc;
  }
  __syncthreads();
  for (int co = threadIdx.x; co < Co; co += 256) {
    float v = 0.f;
    for (int l = 0; l < nwl; ++l) v += red[l * Co + co];
    part[(int64_t)blockIdx.x * Co + co] = v;
  }
}

// the operand split on its own: planes of n bf16 each (the packed weights, once per step; tests)
__global__ void x3_split_kernel(const float4* x, uint2* hi, uint2* mid, uint2* lo, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) split4(x[i], hi[i], mid[i], lo[i], split_consts());
}

// fp32 [n] -> x3-packed: 24 bytes per 4 elements, hi[4] mid[4] lo[4]
__global__ void x3_pack_kernel(const float4* x, uint2* out, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  uint2 h, m, l;
  split4(x[i], h, m, l, split_consts());
  out[3 * i] = h; out[3 * i + 1] = m; out[3 * i + 2] = l;
}

// bias gradient (+ optional packed copy) of a pooled gradient [rows][Co]; `parts` = kBiasParts * Co floats of workspace
static int launch_bias_grad(const float* dp, const uint8_t* am, float* dbias, void* packed, int64_t rows, int Co, float* parts,
                            hipStream_t s) {
  const int nparts = rows < kBiasParts ? (int)rows : kBiasParts;
  const int64_t per = (rows + nparts - 1) / nparts;
  const size_t lds = (size_t)(256 / (Co / 4)) * Co * 4;
  if (packed)
    hipLaunchKernelGGL(conv_bias_grad_kernel<true>, dim3(nparts), dim3(256), lds, s, dp, am, parts, rows, Co, per,
                       static_cast<uint2*>(packed));
  else
    hipLaunchKernelGGL(conv_bias_grad_kernel<false>, dim3(nparts), dim3(256), lds, s, dp, am, parts, rows, Co, per,
                       (uint2*)nullptr);
  int rc = check_hip(hipGetLastError(), "conv_bias_grad launch");
  if (rc) return rc;
  hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3((Co + 31) / 32), dim3(256), 0, s, parts, dbias, nparts, Co);
  return check_hip(hipGetLastError(), "wgrad_bias_reduce launch");
}

static bool x3_conv_ok(int CiP, int Co, int Wp) { return CiP % BK == 0 && Co % BK == 0 && 2 * Wp >= BK; }
// images per launch: as batch_chunk, with the tensors at 6 bytes per element (the packed forms; also used for fp32
// tensors so that the forward / dgrad / wgrad / workspace computations agree whatever the form)
static int x3_chunk(int B, int H, int W, int CiP, int Co, int stride) {
  return batch_chunk(B, H, W, CiP / 2 * 3, Co / 2 * 3, stride);
}

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

int vqa_x3_pack(const float* x, void* out, int64_t n, vqa_stream_t stream) {
  VQA_REQUIRE(x && out && n > 0 && n % 4 == 0, "vqa_x3_pack: bad args (n=%lld must be a multiple of 4)", (long long)n);
  hipLaunchKernelGGL(x3_pack_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(x), static_cast<uint2*>(out), n / 4);
  return check_hip(hipGetLastError(), "x3_pack launch");
}

int64_t vqa_x3_pack_pooled_grad_workspace_bytes(int Co) { return (int64_t)kBiasParts * Co * 4; }

int vqa_x3_pack_pooled_grad(const float* dpooled, const uint8_t* argmax, void* packed, float* dbias, int64_t windows, int Co,
                            float* workspace, int64_t workspace_bytes, vqa_stream_t stream) {
  VQA_REQUIRE(dpooled && argmax && packed && dbias && workspace && windows > 0, "vqa_x3_pack_pooled_grad: null pointer");
  VQA_REQUIRE(Co % 4 == 0 && Co > 0 && Co <= 1024, "vqa_x3_pack_pooled_grad: Co=%d (multiple of 4, <= 1024)", Co);
  VQA_REQUIRE(workspace_bytes >= vqa_x3_pack_pooled_grad_workspace_bytes(Co), "vqa_x3_pack_pooled_grad: workspace too small");
  return launch_bias_grad(dpooled, argmax, dbias, packed, windows, Co, workspace, (hipStream_t)stream);
}

int vqa_conv3x3_x3_supported(int H, int W, int CiP, int Co, int stride) {
  const ConvGeom g = make_geom(1, H, W, CiP, Co, stride);
  return (g.Hp > 0 && g.Wp > 0 && x3_conv_ok(CiP, Co, g.Wp)) ? 1 : 0;
}

int vqa_conv3x3_relu_pool_fwd_x3(const void* x, int x_packed, const void* wf, const float* bias, void* pooled,
                                 int pooled_packed, uint8_t* argmax, int B, int H, int W, int CiP, int Co, int stride,
                                 int tag, vqa_stream_t stream) {
  VQA_REQUIRE(x && wf && bias && pooled && argmax && B > 0, "vqa_conv3x3_relu_pool_fwd_x3: null pointer");
  VQA_REQUIRE(CiP % BK == 0, "vqa_conv3x3_relu_pool_fwd_x3: CiP=%d must be a multiple of %d", CiP, BK);
  const int chunk = x3_chunk(B, H, W, CiP, Co, stride);
  using AF = ConvFwdA<CfgX::NVA, CfgX::LT, true>;
  using AFn = ConvFwdA<CfgXn::NVA, CfgXn::LT, true>;
  using AP = ConvFwdAx<CfgX::NVA, CfgX::LT>;
  using APn = ConvFwdAx<CfgXn::NVA, CfgXn::LT>;
  const char* xb = static_cast<const char*>(x);
  const int esz = x_packed ? 6 : 4;
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
    const void* xc = xb + xo * esz;
    void* pc = static_cast<char*>(pooled) + po * (pooled_packed ? 6 : 4);
    hipStream_t st = (hipStream_t)stream;
#define X3_FWD(CFG, AL, OP) launch_fwd_x3<CFG, AL, OP>(xc, wf, bias, pc, argmax + po, g, st)
    if (pooled_packed)
      rc = x_packed ? (Co > 64 ? X3_FWD(CfgX, AP, true) : X3_FWD(CfgXn, APn, true))
                    : (Co > 64 ? X3_FWD(CfgX, AF, true) : X3_FWD(CfgXn, AFn, true));
    else
      rc = x_packed ? (Co > 64 ? X3_FWD(CfgX, AP, false) : X3_FWD(CfgXn, APn, false))
                    : (Co > 64 ? X3_FWD(CfgX, AF, false) : X3_FWD(CfgXn, AFn, false));
#undef X3_FWD
    if (rc) return rc;
  }
  return VQA_OK;
}

int vqa_conv3x3_dgrad_x3(const void* dpooled, int dp_packed, const uint8_t* argmax, const void* wd, float* dx, int B,
                         int H, int W, int CiP, int Co, int stride, int tag, vqa_stream_t stream) {
  VQA_REQUIRE(dpooled && argmax && wd && dx && B > 0, "vqa_conv3x3_dgrad_x3: null pointer");
  VQA_REQUIRE(Co % BK == 0, "vqa_conv3x3_dgrad_x3: Co=%d must be a multiple of %d", Co, BK);
  const int chunk = x3_chunk(B, H, W, CiP, Co, stride);
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
    const void* dpc = static_cast<const char*>(dpooled) + po * (dp_packed ? 6 : 4);
    hipStream_t st = (hipStream_t)stream;
    if (dp_packed)
      rc = CiP > 64 ? launch_dgrad_x3<CfgXd, ConvDgradAx<CfgXd::NVA, CfgXd::LT>>(dpc, argmax + po, wd, dx + xo, g, st)
                    : launch_dgrad_x3<CfgXnd, ConvDgradAx<CfgXnd::NVA, CfgXnd::LT>>(dpc, argmax + po, wd, dx + xo, g, st);
    else
      rc = CiP > 64 ? launch_dgrad_x3<CfgXd, ConvDgradA<CfgXd::NVA, CfgXd::LT, true>>(dpc, argmax + po, wd, dx + xo, g, st)
                    : launch_dgrad_x3<CfgXnd, ConvDgradA<CfgXnd::NVA, CfgXnd::LT, true>>(dpc, argmax + po, wd, dx + xo, g, st);
    if (rc) return rc;
  }
  return VQA_OK;
}

int64_t vqa_conv3x3_wgrad_x3_workspace_bytes(int B, int H, int W, int CiP, int Co, int stride) {
  const ConvGeom g1 = make_geom(1, H, W, CiP, Co, stride);
  if (g1.Hp <= 0 || g1.Wp <= 0 || B <= 0 || CiP % BK) return 0;
  const int chunk = x3_chunk(B, H, W, CiP, Co, stride);
  if (chunk <= 0) return 0;
  int64_t parts = 0;
  for (int b0 = 0; b0 < B; b0 += chunk)
    parts += plan_wgrad_x3(make_geom(B - b0 < chunk ? B - b0 : chunk, H, W, CiP, Co, stride)).splits;
  return parts * (int64_t)9 * CiP * Co * 4 + (int64_t)kBiasParts * Co * 4;
}

int vqa_conv3x3_wgrad_x3(const void* x, int x_packed, const float* dpooled, const void* dpooled_packed,
                         const uint8_t* argmax, float* dw, float* dbias, int B, int H, int W, int CiP, int Ci, int Co,
                         int stride, float* workspace, int64_t workspace_bytes, int tag, vqa_stream_t stream) {
  VQA_REQUIRE(x && dpooled && argmax && dw && workspace && B > 0, "vqa_conv3x3_wgrad_x3: null pointer");
  VQA_REQUIRE(Ci >= 1 && Ci <= CiP && Co <= 1024, "vqa_conv3x3_wgrad_x3: Ci=%d CiP=%d Co=%d (Co <= 1024)", Ci, CiP, Co);
  const ConvGeom g1 = make_geom(1, H, W, CiP, Co, stride);
  VQA_REQUIRE(g1.Hp > 0 && g1.Wp > 0 && x3_conv_ok(CiP, Co, g1.Wp),
              "vqa_conv3x3_wgrad_x3: needs CiP, Co multiples of %d and 2*Wp >= %d (CiP=%d Co=%d Wp=%d)", BK, BK, CiP, Co, g1.Wp);
  const int chunk = x3_chunk(B, H, W, CiP, Co, stride);
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
  float* const bias_parts = workspace + (int64_t)parts * KI * Co;
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
    using SL = SmemLayoutX<CfgXw, false, false>;
    WgradGeom wg{g.H, g.W, g.CiP, g.Hp, g.Wp, g.Co, g.stride, p.Mtot};
    const int64_t po = (int64_t)b0 * g1.Hp * g1.Wp * Co;
    const char* xc = static_cast<const char*>(x) + (int64_t)b0 * H * W * CiP * (x_packed ? 6 : 4);
    const dim3 grid(p.tiles_m * p.tiles_n * p.splits);
    float* slab = workspace + (int64_t)done * KI * Co;
    using AF = WgradA<CfgXw::NVA, CfgXw::LT, true>;
    using AP = WgradAx<CfgXw::NVA, CfgXw::LT>;
    using BF = WgradB<CfgXw::NVB, CfgXw::LT, true>;
    using BP = WgradBx<CfgXw::NVB, CfgXw::LT>;
#define X3_WGRAD(AL, BL, XPTR, DPTR)                                                                                   \
    {                                                                                                                  \
      typename AL::Params pa{XPTR, wg, p.KI};                                                                          \
      typename BL::Params pb{DPTR, argmax + po, wg};                                                                   \
      auto kern = conv_wgrad_x3_kernel<CfgXw, AL, BL>;                                                                  \
      rc = set_smem(kern, SL::BYTES, "attr(conv_wgrad_x3)");                                                           \
      if (rc) return rc;                                                                                               \
      hipLaunchKernelGGL(kern, grid, dim3(CfgXw::THREADS), SL::BYTES, s, pa, pb, slab, p.tiles_m, p.tiles_n, p.nk,      \
                         p.ks_per_split);                                                                              \
    }
    const float* xf = reinterpret_cast<const float*>(xc);
    const float* dpf = dpooled + po;
    const void* dpp = dpooled_packed ? static_cast<const char*>(dpooled_packed) + po * 6 : nullptr;
    if (x_packed && dpp) X3_WGRAD(AP, BP, xc, dpp)
    else if (x_packed) X3_WGRAD(AP, BF, xc, dpf)
    else if (dpp) X3_WGRAD(AF, BP, xf, dpp)
    else X3_WGRAD(AF, BF, xf, dpf)
#undef X3_WGRAD
    rc = check_hip(hipGetLastError(), "conv_wgrad_x3 launch");
    if (rc) return rc;
    done += p.splits;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((KI * Co + 63) / 64), dim3(256), 0, s, workspace, dw, parts, KI, CiP, Ci,
                     Co);
  rc = check_hip(hipGetLastError(), "wgrad_reduce launch");
  if (rc) return rc;
  // bias gradient = sum of the pooled gradient over the windows whose ReLU was alive (arg-max byte != 4); dbias == NULL:
  // the caller already has it (vqa_x3_pack_pooled_grad)
  if (!dbias) return VQA_OK;
  return launch_bias_grad(dpooled, argmax, dbias, nullptr, (int64_t)B * g1.Hp * g1.Wp, Co, bias_parts, s);
}

}  // extern "C"
