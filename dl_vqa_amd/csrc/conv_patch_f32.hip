// fp32 "patch" backward-data convolution for gfx950 (the fp32 headline path; models/model.py:80-82, autograd of Conv2d +
// ReLU + MaxPool2d): dX of a 3x3 stride-1 block from its pooled gradient and arg-max bytes.  VERDICT r2 item 4.
//
// Why.  The implicit-GEMM backward-data kernel (conv.hip) rebuilds the routed pre-pool gradient in its loader waves once per
// TAP (nine times per element) and, for the 64-channel block, shares an A tile among only 64 output columns: 74-81 % of the
// fp32 MFMA peak where the forward kernel reaches 88 %, and 81 % at best with the routing compiled out
// (profiles/r03_kbench_noroute.txt).  Here -- the structure of conv_patch_bf16.hip -- a persistent workgroup BUILDS the
// pre-pool gradient of one 8-channel K-slice in LDS once and takes the nine taps as nine shifted fragment reads.
//
// Flattened tiles.  At 224 x 224 the maps are 111 and 54 pixels wide; 2-D tiles of 16 / 32 pixels would lose 14-29 % to
// edges.  A tile is instead a SEGMENT of SEG consecutive positions of the image's output in row-major order with the padded
// row length W' = W + 2 (the pre-pool gradient, offset by the 2-pixel border, is exactly W' wide): output m = y * W' + x
// reads flat[m + ky * W' + kx], so a tap is still ONE constant shift of the LDS patch, an MFMA A fragment is 32 consecutive
// floats (conflict-free), and the only waste is the 2 garbage columns per row (x >= W) and the last segment of an image:
// 92.6 % (111 x 111) and 94.9 % (54 x 54) useful.
//   * workgroup = 8 waves, one per CU; a wave owns 128 positions x 64 channels = 4 x 2 accumulators of
//     v_mfma_f32_32x32x2_f32 (exact fp32); WM = 8 waves along the segment: SEG = 1024 positions x 64 channels, or 4 x 2:
//     512 x 128;
//   * a stage = the patch [8 channels][SEG + 2 W' + 2] fp32, channel-planar, + the slice's weights for all nine taps
//     [tap][k][n] fp32, pre-packed flipped and transposed (LDS-DMA, 1 KiB per piece, spread behind the taps): 288 MFMAs of 64
//     cycles per wave and stage -- one s_barrier per 18 k cycles;
//   * the patch is ROUTED: thread t takes one 2 x 2 window of the EXTENDED window grid (border and uncovered pixels are
//     windows without data, so every patch position is written by exactly one thread -- no zero fill), loads 8 channels of dP +
//     arg-max for stage q + 2 during stage q and writes the window's four pixels per channel during stage q + 1;
//   * epilogue: lane (channel, h) holds 16 positions of one channel per accumulator: 32 lanes write one pixel's 128 bytes.
#include "bf16_core.hpp"
#include <stdlib.h>

namespace vqa {

constexpr int PF_WMAX = 256;                       // input maps up to 256 wide (W' = 258)

template <int WM_>
struct PfCfg {
  static constexpr int WM = WM_, WN = 8 / WM_;
  static constexpr int SEG = 128 * WM;                                       // output positions per tile
  static constexpr int NSLAB = 64 * WN;
  static constexpr int PLANE = SEG + 2 * (PF_WMAX + 2) + 8;                  // floats per channel plane (>= SEG + 2 W' + 2)
  static constexpr int PATCH_BYTES = ((8 * PLANE * 4 + 1023) / 1024) * 1024;
  static constexpr int W_BYTES = 9 * 8 * NSLAB * 4;
  static constexpr int W_INSTR = W_BYTES / 1024;                             // 18 / 36
  static constexpr int WK = (W_INSTR + 7) / 8;                               // weight pieces per wave and stage: 3 / 5
  static constexpr int LDS = 2 * (PATCH_BYTES + W_BYTES);
  static constexpr int RT_K = 2;                                             // routed windows per thread and stage, at most
};

struct PfParams {
  const char* dp;       // pooled gradient fp32 NHWC [B][Hq][Wq][Cin]
  const char* am;       // arg-max bytes NHWC [B][Hq][Wq][Cin]
  const char* wimg;     // packed weights [nslabs][Cin/8][9][8][NSLAB] fp32 (flipped, transposed)
  float* out;           // dX fp32 NHWC [B][H][W][N]
  int B, Cin, N, H, W, Wp2, Hq, Wq;        // Wp2 = W + 2
  int EW;               // extended windows per row: (Wp2 + 1) / 2
  int segs, nunits;     // segments per image, B * segs
  int nslabs, nslices;
  int dbg;              // timing experiments (VQA_PCONVF_DBG): 1 = no routing after the prologue, 2 = no weight DMA after the prologue, 4 = no epilogue stores
};

typedef unsigned int pf_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ pf_rsrc_t pf_rsrc(const void* base, uint32_t bytes = 0xffff0000u) {
  const uint64_t a = (uint64_t)base;
  pf_rsrc_t r;
  r.x = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r.y = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
// 16 bytes per lane global -> LDS (see conv_patch_bf16.hip: inline asm keeps hipcc's wait insertion out of the way)
__device__ __forceinline__ void pf_dma16(pf_rsrc_t r, const void* lds_dst, uint32_t voff, uint32_t soff) {
  const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds_dst);
  uint32_t keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(m0v), "v"(voff), "s"(r), "s"(soff) : "memory");
}

template <int WMv>
__global__ __launch_bounds__(512, 2) void pconvf_dgrad_kernel(const PfParams P) {
  using C = PfCfg<WMv>;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int r = lane & 31, h = lane >> 5;

  const int bid = blockIdx.x;
  const int slab = (bid >> 3) % P.nslabs;
  const int stream = ((bid >> 3) / P.nslabs) * 8 + (bid & 7);
  const int nstreams = gridDim.x / P.nslabs;
  const int my_tiles = stream < P.nunits ? (P.nunits - stream + nstreams - 1) / nstreams : 0;
  if (my_tiles == 0) return;                         // uniform over the workgroup
  const int n_slab0 = slab * C::NSLAB;
  const int PLEN = C::SEG + 2 * P.Wp2 + 2;           // patch positions of a tile

  // fragment bases (bytes): A = patch[2 ks + h][wm*128 + 32 i + r + ky W' + kx], B = w[tap][2 ks + h][wn*64 + 32 j + r]
  const uint32_t a_lane = (uint32_t)((h * C::PLANE + wm * 128 + r) * 4);
  const uint32_t b_lane = (uint32_t)(2 * C::PATCH_BYTES + (h * C::NSLAB + wn * 64 + r) * 4);

  // ---- weights by DMA: piece i = wave + 8 n of the stage's W_INSTR linear KiB
  const pf_rsrc_t wrs = pf_rsrc(P.wimg);
  const uint32_t wlane = (uint32_t)lane * 16u;
  uint32_t ns_wsrc = 0;
  int ns_buf = 0;
  bool ns_on = false;
  auto issue_piece = [&](int n) {
    if (!ns_on) return;
    const int i = wave + 8 * n;
    if (n < C::WK && i < C::W_INSTR)
      pf_dma16(wrs, smem + 2 * C::PATCH_BYTES + ns_buf * C::W_BYTES + i * 1024, wlane, ns_wsrc + i * 1024);
  };

  // ---- routed patch.  Extended window grid: window (er, e) covers the padded pixels (2 er + dy, 2 e + dx); it has data iff
  // qy = er - 1 in [0, Hq) and qx = e - 1 in [0, Wq).  Task t of a tile = window (er0 + t / EW, t % EW), er0 = first patch row / 2.
  int t_er[C::RT_K], t_e[C::RT_K];
#pragma unroll
  for (int k = 0; k < C::RT_K; ++k) {
    const int t = k * 512 + (int)threadIdx.x;
    t_er[k] = t / P.EW;
    t_e[k] = t - t_er[k] * P.EW;
  }
  float4 rt_d0[C::RT_K], rt_d1[C::RT_K];
  uint2 rt_a[C::RT_K];
  int rt_unit = stream, rt_slice = 0, rt_count = 0;         // the stage being fetched
  const int rt_total = my_tiles * P.nslices;
  auto rt_load = [&]() {           // always issued (past the stream: out-of-range offsets, zeros)
    const bool on = rt_count < rt_total;
    const int img = rt_unit / P.segs, seg = rt_unit - img * P.segs;
    const int er0 = (seg * C::SEG / P.Wp2) >> 1;
    const int64_t ibase = (int64_t)img * P.Hq * P.Wq * P.Cin;
    const uint32_t ibytes = (uint32_t)(P.Hq * P.Wq * P.Cin);
    const __amdgpu_buffer_rsrc_t rd = buf_rsrc(P.dp + ibase * 4, ibytes * 4u), ra = buf_rsrc(P.am + ibase, ibytes);
#pragma unroll
    for (int k = 0; k < C::RT_K; ++k) {
      const int qy = er0 + t_er[k] - 1, qx = t_e[k] - 1;
      const bool ok = on && (unsigned)qy < (unsigned)P.Hq && (unsigned)qx < (unsigned)P.Wq;
      const uint32_t o = (uint32_t)((qy * P.Wq + qx) * P.Cin + 8 * rt_slice);
      rt_d0[k] = buf_load16(rd, ok ? o * 4u : BUF_OOB);
      rt_d1[k] = buf_load16(rd, ok ? o * 4u + 16u : BUF_OOB);
      rt_a[k] = buf_load8(ra, ok ? o : BUF_OOB);
    }
    ++rt_count;
    if (++rt_slice == P.nslices) { rt_slice = 0; rt_unit += nstreams; }
  };
  // the registers -> the four pixels of each task's window, in patch buffer `buf`, for the tile whose first position is p0
  auto rt_route = [&](int buf, int p0) {
    char* const dst = smem + buf * C::PATCH_BYTES;
    const int er0 = (p0 / P.Wp2) >> 1;
#pragma unroll
    for (int k = 0; k < C::RT_K; ++k) {
      const float d[8] = {rt_d0[k].x, rt_d0[k].y, rt_d0[k].z, rt_d0[k].w, rt_d1[k].x, rt_d1[k].y, rt_d1[k].z, rt_d1[k].w};
      const int yy = 2 * (er0 + t_er[k]), xx = 2 * t_e[k];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pl = (yy + (j >> 1)) * P.Wp2 + xx + (j & 1) - p0;       // position in the patch
        if (xx + (j & 1) < P.Wp2 && pl >= 0 && pl < PLEN) {
#pragma unroll
          for (int ch = 0; ch < 8; ++ch) {
            const uint32_t code = ((ch < 4 ? rt_a[k].x : rt_a[k].y) >> (8 * (ch & 3))) & 0xffu;
            *reinterpret_cast<float*>(dst + (ch * C::PLANE + pl) * 4) = code == (uint32_t)j ? d[ch] : 0.f;
          }
        }
      }
    }
  };

  f32x16 acc[4][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  };
  zero_acc();

  // tap shifts in bytes
  uint32_t toff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) toff[t] = (uint32_t)(((t / 3) * P.Wp2 + t % 3) * 4);

  // prologue: stage 0's weights by DMA, its patch routed; stage 1's operands in registers
  ns_wsrc = (uint32_t)((slab * P.nslices + 0) * C::W_BYTES);
  ns_buf = 0;
  ns_on = true;
#pragma unroll
  for (int n = 0; n < C::WK; ++n) issue_piece(n);
  rt_load();
  rt_route(0, (stream % P.segs) * C::SEG);
  rt_load();
  int buf = 0, gstage = 0;
  int unit = stream;
  for (int k = 0; k < my_tiles; ++k, unit += nstreams) {
    const int img = unit / P.segs, seg = unit - img * P.segs;
    const int p0 = seg * C::SEG;
    for (int slice = 0; slice < P.nslices; ++slice, ++gstage) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's DMA pieces and patch writes are done
      __builtin_amdgcn_s_barrier();
      const bool last_slice = slice + 1 == P.nslices;
      {
        ns_wsrc = (uint32_t)((slab * P.nslices + (last_slice ? 0 : slice + 1)) * C::W_BYTES);
        ns_buf = buf ^ 1;
        ns_on = gstage + 1 < rt_total;
      }
      // first position of the NEXT stage's tile (its patch is routed during this stage)
      const int np0 = last_slice ? ((unit + nstreams) % P.segs) * C::SEG : p0;
      const char* const pa = smem + buf * C::PATCH_BYTES + a_lane;
      const char* const pb = smem + buf * C::W_BYTES + b_lane;
      float a[2][4], b[2][2];
      auto fetch = [&](int t, int ks, int set) {
        const char* const pat = pa + toff[t];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[set][i] = *reinterpret_cast<const float*>(pat + ((2 * ks) * C::PLANE + 32 * i) * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          b[set][j] = *reinterpret_cast<const float*>(pb + ((t * 8 + 2 * ks) * C::NSLAB + 32 * j) * 4);
      };
      fetch(0, 0, 0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int q = 4 * t + ks;
          // the next k-step's six fragments are READ here, a whole k-step (8 MFMAs = 512 cycles) before their use; the
          // scheduling barriers keep hipcc from sinking the reads next to their consumers (it did: ds_read, s_waitcnt
          // lgkmcnt(0), four MFMAs -- the LDS latency in front of every half k-step, 84 % MFMA utilisation)
          if (q + 1 < 36) fetch((q + 1) >> 2, (q + 1) & 3, (q + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][i], b[q & 1][j], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t == 0) { if (ns_on && !(P.dbg & 1)) rt_route(buf ^ 1, np0); }  // the next stage's patch (registers loaded a stage ago)
        else if (t == 1) rt_load();                          // the stage after that
        else if (t < 2 + C::WK) { if (!(P.dbg & 2)) issue_piece(t - 2); }   // the next stage's weights
        __builtin_amdgcn_sched_barrier(0);
      }
      buf ^= 1;
    }
    // ---- epilogue: position m = p0 + wm*128 + 32 i + (e & 3) + 8 (e >> 2) + 4 h -> (y, x) = (m / W', m % W'), valid for x < W, y < H
    const int64_t obase = (int64_t)img * P.H * P.W * P.N + n_slab0 + wn * 64;
    const __amdgpu_buffer_rsrc_t ro = buf_rsrc(P.out + obase);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = p0 + wm * 128 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int y = m / P.Wp2, x = m - y * P.Wp2;
        const bool ok = x < P.W && y < P.H && !(P.dbg & 4);
        const uint32_t vo = ok ? (uint32_t)(((y * P.W + x) * P.N + r) * 4) : BUF_OOB;
#pragma unroll
        for (int j = 0; j < 2; ++j) buf_store4(ro, acc[i][j][e], vo, (uint32_t)(32 * j * 4));
      }
    zero_acc();
  }
}

// packed image [slab][slice][tap][k][n] (fp32): w[co = 8 slice + k][ci = slab*nslab + n][2 - ky][2 - kx]
__global__ void pconvf_pack_kernel(const float* w, float* img, int Co, int Ci, int nslab) {
  const int64_t total = (int64_t)9 * Ci * Co;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int n = (int)(idx % nslab);
  int64_t t = idx / nslab;
  const int k = (int)(t % 8); t /= 8;
  const int tap = (int)(t % 9); t /= 9;
  const int nslices = Co / 8;
  const int slice = (int)(t % nslices);
  const int slab = (int)(t / nslices);
  const int ky = tap / 3, kx = tap - 3 * ky;
  const int co = 8 * slice + k, ci = slab * nslab + n;
  img[idx] = w[((int64_t)co * Ci + ci) * 9 + (2 - ky) * 3 + (2 - kx)];
}

static int pf_nslab(int N) { return N % 128 == 0 ? 128 : 64; }

template <int WMv>
static int pf_launch(PfParams P, hipStream_t s) {
  using C = PfCfg<WMv>;
  P.segs = (P.H * P.Wp2 + C::SEG - 1) / C::SEG;
  P.nunits = P.B * P.segs;
  auto kern = pconvf_dgrad_kernel<WMv>;
  int rc = ensure_dyn_smem(reinterpret_cast<const void*>(kern), C::LDS, "attr(pconvf_dgrad)");
  if (rc) return rc;
  int grid = 256;
  grid -= grid % (8 * P.nslabs);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), C::LDS, s, P);
  return check_hip(hipGetLastError(), "pconvf_dgrad launch");
}

}  // namespace vqa

using namespace vqa;

extern "C" {

int vqa_pconvf_supported(int H, int W, int Ci, int Co, int stride) {
  if (stride != 1 || H < 4 || W < 4 || W > PF_WMAX || Ci % 64 || Co % 8 || Ci <= 0 || Co <= 0 || Ci > 4096 || Co > 4096) return 0;
  if (256 % (8 * (Ci / pf_nslab(Ci)))) return 0;
  if ((int64_t)H * W * (Ci > Co ? Ci : Co) * 4 >= (1LL << 31)) return 0;      // 32-bit byte offsets inside one image
  // the routed windows of a tile must fit 2 x 512 threads: rows spanned / 2 + 2 extended rows of (W + 3) / 2 windows
  const int Wp2 = W + 2, seg = pf_nslab(Ci) == 128 ? 512 : 1024;
  const int rows = (seg + 2 * Wp2 + 2) / Wp2 + 2;
  if ((rows / 2 + 2) * ((Wp2 + 1) / 2) > 1024) return 0;
  return 1;
}

int64_t vqa_pconvf_weights_bytes(int Ci, int Co) { return (int64_t)9 * Ci * Co * 4; }

int vqa_pconvf_pack_weights(const float* w, float* wd_img, int Co, int Ci, vqa_stream_t stream) {
  VQA_REQUIRE(w && wd_img && Ci % 64 == 0 && Co % 8 == 0, "vqa_pconvf_pack_weights: bad args Co=%d Ci=%d", Co, Ci);
  const int64_t total = (int64_t)9 * Ci * Co;
  hipLaunchKernelGGL(pconvf_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, wd_img, Co, Ci,
                     pf_nslab(Ci));
  return check_hip(hipGetLastError(), "pconvf_pack launch");
}

int vqa_pconvf_dgrad(const float* dpooled, const uint8_t* argmax, const float* wd_img, float* dx, int B, int H, int W, int Ci, int Co,
                     int tag, vqa_stream_t stream) {
  VQA_REQUIRE(dpooled && argmax && wd_img && dx && B > 0, "vqa_pconvf_dgrad: null pointer");
  VQA_REQUIRE(vqa_pconvf_supported(H, W, Ci, Co, 1), "vqa_pconvf_dgrad: unsupported shape H=%d W=%d Ci=%d Co=%d", H, W, Ci, Co);
  PfParams P{};
  P.dp = reinterpret_cast<const char*>(dpooled);
  P.am = reinterpret_cast<const char*>(argmax);
  P.wimg = reinterpret_cast<const char*>(wd_img);
  P.out = dx;
  P.B = B; P.Cin = Co; P.N = Ci; P.H = H; P.W = W; P.Wp2 = W + 2;
  P.Hq = (H - 2) / 2; P.Wq = (W - 2) / 2;
  VQA_REQUIRE(P.Hq > 0 && P.Wq > 0, "vqa_pconvf_dgrad: image too small");
  P.EW = (P.Wp2 + 1) / 2;
  const int nslab = pf_nslab(Ci);
  P.nslabs = Ci / nslab;
  P.nslices = Co / 8;
  {
    const char* e = getenv("VQA_PCONVF_DBG");
    P.dbg = e ? atoi(e) : 0;
  }
  hipStream_t s = (hipStream_t)stream;
  set_launch_tag(tag);
  ProfScope prof(VQA_K_CONV_DGRAD, s);
  return nslab == 128 ? pf_launch<4>(P, s) : pf_launch<8>(P, s);
}

}  // extern "C"
