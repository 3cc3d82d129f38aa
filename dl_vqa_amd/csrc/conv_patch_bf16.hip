// bf16 "patch" convolutions for gfx950 (BASELINE configs[3]; models/model.py:80-82 and its autograd): 3x3, stride 1.
//
// Why a second formulation.  The implicit-GEMM kernels of conv_bf16.inc re-fetch every input pixel once per tap: at the
// bf16 MFMA rate that needs 32-64 B/clk/CU of L2->LDS traffic for the 64/128-channel blocks, the CU takes in ~30
// (MI355X_MICROARCH.md, "Indexed rows"), and the kernels sat at 15-37 % of the bf16 peak (round 2).  Here a workgroup
// keeps an input PATCH in LDS and takes the nine taps as nine shifted fragment reads of it, so a pixel is fetched once
// per workgroup (7-13 B/clk/CU at peak):
//   * workgroup = 8 waves (2 per SIMD), persistent (one per CU, walks a stream of output tiles);
//     a wave owns 4 (rows) x 32 (columns) output pixels x 64 output channels = 4 x 2 accumulators of 32x32 (128 VGPRs);
//     WM = 4 wave rows x 2 wave columns: tile 16 x 32 pixels x 128 channels, or WM = 8 x 1: 32 x 32 pixels x 64 channels;
//   * K is walked in SLICES of 16 input channels (one v_mfma_f32_32x32x16_bf16 k-step per tap): a stage = the
//     (TY+2) x 40-pixel patch of one slice (32 bytes per pixel) + the slice's weights for all nine taps, pre-packed on
//     the device in fragment order (a B fragment is one linear 1-KiB read).  Two stages in LDS (120 KiB); stage q+1 is
//     fetched by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR round trip, no ds_write) while stage q is computed:
//     ONE s_waitcnt vmcnt(0) + ONE s_barrier per stage of 72 MFMAs per wave;
//   * an MFMA row tile = 2 rows x 16 columns of pixels with row index 4 * window + (dy, dx), so the four pixels of a
//     pooling window are four consecutive accumulator registers of one lane (the pool / arg-max epilogue of the
//     implicit-GEMM kernels, unchanged);
//   * LDS image: pixel p of the patch (row stride 40 pixels) holds its two 16-byte channel chunks at slots
//     chunk ^ bit3(p); the swizzle is applied to the per-lane SOURCE address of the DMA (the destination of a DMA is
//     lane-linear) and to the fragment read.  A 16-lane ds_read_b128 group covers 4 windows x 2 x 2 pixels: eight
//     consecutive-mod-8 pixels on two rows whose bit3 differs (40 = 8 mod 16), i.e. all 16 16-byte bank groups once:
//     conflict-free for every tap shift;
//   * HBM layout of everything a patch is cut from (activations between the blocks, the materialised pre-pool gradient):
//     channel-BLOCKED "C16" = [B][C/16][H][W][16] bf16, so that a K-slice of a patch row is one contiguous run and every
//     128-byte line a DMA piece touches is used whole.  With NHWC the 32-byte slice of a 128-/256-/512-byte pixel was a
//     quarter-to-sixteenth of its line, the other slices came 1-15 stages later, and with 32 workgroups per XCD streaming
//     ~350 KB patches the 4 MiB L2 had dropped the line by then: measured TCC hit rate 23-72 %, FETCH_SIZE 2.4-4.7x the
//     tensor (profiles/r03_pconv_nhwc_l2.txt).  Pooled outputs for the L2 norm (fp32) and dX stay NHWC;
//   * backward-data is the SAME kernel: the pre-pool gradient dY is materialised once per layer (pconv_expand_dy: routed
//     by the arg-max bytes, bf16, with a zero border of 2 pixels) and dX = valid conv of that padded map with the
//     flipped, transposed weights; the weight gradient (pconv_wgrad) reads the same dY.
#include "bf16_core.hpp"
#include <stdlib.h>

namespace vqa {

// Timing experiments (tools/kbench_pconv.py --dbg ...): only in a -DVQA_PCONV_DIAG build do the kernels look at VQA_PCONV_DBG
// (1 = no epilogue stores, 2 = no DMA after the first stage, 4 = no MFMA); the shipped kernels carry none of those branches.
#ifdef VQA_PCONV_DIAG
#define PC_DBG(bit) (P.dbg & (bit))
#else
#define PC_DBG(bit) false
#endif

constexpr int PC_RS = 40;   // patch row stride in pixels (34 used; 40 = 8 mod 16 keeps the fragment reads conflict-free)
constexpr int PC_TX = 32;   // tile width in pixels

template <int WM_>
struct PcCfg {
  static constexpr int WM = WM_, WN = 8 / WM_;
  static constexpr int TY = 4 * WM;                  // tile height: 16 / 32 rows
  static constexpr int NSLAB = 64 * WN;              // output channels of a workgroup: 128 / 64
  static constexpr int NT = NSLAB / 32;
  static constexpr int PROWS = TY + 2;
  static constexpr int PATCH_BYTES = ((PROWS * PC_RS * 32 + 1023) / 1024) * 1024;
  static constexpr int PATCH_INSTR = PATCH_BYTES / 1024;
  static constexpr int PK = (PATCH_INSTR + 7) / 8;   // patch DMA pieces per wave (piece i = wave + 8 k)
  static constexpr int W_BYTES = 9 * NT * 1024;
  static constexpr int W_INSTR = 9 * NT;
  static constexpr int WK = (W_INSTR + 7) / 8;
  static constexpr int NP = PK + WK;                 // pieces a wave issues per stage: one behind each tap's MFMAs (two behind tap 0 when NP = 9)
  static_assert(NP <= 9, "at most nine pieces per wave and stage");
  static constexpr int LDS = 2 * (PATCH_BYTES + W_BYTES);
  static constexpr int SCR = 4608;                   // wave-private epilogue scratch (>= 4 KiB: 32 pixels x 64 channels bf16)
  static constexpr int LDS_ALL = LDS + 8 * SCR;
  // global stores a wave issues per tile epilogue (the counted vmcnt wait of the stage that follows)
  template <int EPI> static constexpr int nstores() { return EPI == 0 ? 8 : EPI == 1 ? 12 : 16; }   // EPI 3 does not count
};

struct PcParams {
  const char* x;        // input map, C16: [B][Cin/16][H][W][16] bf16
  const char* x_end;    // one past its last byte
  const char* wimg;     // packed weights [nslabs][Cin/16][9][NT][64 lanes][8] bf16
  const float* bias;    // [N] (forward) or null
  void* out;            // EPI 0: pooled bf16 C16 [B][N/16][Hp][Wp][16]; EPI 1: pooled fp32 [B][Hp][Wp][N]; plain: [B][Hc][Wc][N]
  uint8_t* amax;        // forward: arg-max bytes, C16 [B][N/16][Hp][Wp][16]
  const char* am_in;    // routed (backward-data) source: arg-max bytes of the block, C16 [B][Cin/16][Hq][Wq][16]; x = dP, C16 bf16
  int Hq, Wq;           // routed source: pooled map size
  int B, H, W, Cin, N;  // input dims, output channels
  int Hc, Wc;           // computed output extent (forward: 2*Hp x 2*Wp; plain: H-2 x W-2)
  int Hp, Wp;           // forward only
  int tiles_y, tiles_x, ntiles;   // spatial tiles per image / in all
  int nslabs, nslices;
  int dbg;              // timing experiments only (VQA_PCONV_DBG): 1 = no epilogue stores, 2 = no DMA after the first stage, 4 = no MFMA
};

typedef __attribute__((address_space(3))) void lds_void;

// LDS-DMA in inline asm.  With the builtin (raw_ptr_buffer_load_lds) hipcc tracks the transfer as a pending LDS write and
// puts s_waitcnt vmcnt(0) in front of LDS accesses it cannot tell apart from the destination -- the first ds_read_b64_tr_b16
// of the weight-gradient stage and the epilogue's scratch accesses here -- i.e. the wave that had just issued the NEXT stage's
// pieces waited for them to land before computing the current one.  The asm form is invisible to that pass; the kernels
// order DMA against LDS reads themselves (one counted s_waitcnt vmcnt + s_barrier per stage).  16 bytes per lane, LDS
// destination = wave-uniform address + 16 * lane (M0), source = descriptor base + voff (per lane) + soff (scalar).
typedef unsigned int pc_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ pc_rsrc_t pc_rsrc(const void* base, uint32_t bytes = 0xffff0000u) {
  const uint64_t a = (uint64_t)base;
  pc_rsrc_t r;
  r.x = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r.y = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
__device__ __forceinline__ void lds_dma16(pc_rsrc_t r, const void* lds_dst, uint32_t voff, uint32_t soff) {
  const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds_dst);
  uint32_t keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(m0v), "v"(voff), "s"(r), "s"(soff) : "memory");
}

// position of a tile in the workgroup's stream; stepping by `nstreams` tiles without divisions (all scalar)
struct PcTile {
  int img, ty, tx;
  __device__ __forceinline__ void decode(int s, int tiles_y, int tiles_x) {
    const int per = tiles_y * tiles_x;
    img = s / per;
    const int rem = s - img * per;
    ty = rem / tiles_x;
    tx = rem - ty * tiles_x;
  }
  __device__ __forceinline__ void advance(const PcTile& d, int tiles_y, int tiles_x) {
    tx += d.tx;
    if (tx >= tiles_x) { tx -= tiles_x; ++ty; }
    ty += d.ty;
    if (ty >= tiles_y) { ty -= tiles_y; ++img; }
    img += d.img;
  }
};

// the 8 bf16 of d whose arg-max byte (id) equals j; zeros elsewhere
__device__ __forceinline__ float4 pc_route8(float4 d, uint2 id, uint32_t j) {
  const uint32_t jj = j * 0x01010101u;
  auto bytemask = [&](uint32_t w) {
    const uint32_t q = w ^ jj;                                   // 0 where the byte equals j (bytes are 0..4, j 0..3)
    return (((0x80808080u - q) & 0x80808080u) >> 7) * 0xffu;
  };
  const uint32_t m0 = bytemask(id.x), m1 = bytemask(id.y);
  float4 v;
  v.x = __uint_as_float(__float_as_uint(d.x) & __builtin_amdgcn_perm(0u, m0, 0x01010000u));
  v.y = __uint_as_float(__float_as_uint(d.y) & __builtin_amdgcn_perm(0u, m0, 0x03030202u));
  v.z = __uint_as_float(__float_as_uint(d.z) & __builtin_amdgcn_perm(0u, m1, 0x01010000u));
  v.w = __uint_as_float(__float_as_uint(d.w) & __builtin_amdgcn_perm(0u, m1, 0x03030202u));
  return v;
}

// EPI 0: bias + ReLU + 2x2 max-pool + arg-max, pooled stored as bf16 C16; 1: the same, pooled fp32 NHWC; 2: plain bf16 NHWC
// store; 3: plain fp32 NHWC store (tests); 4: plain bf16 C16 store
// Epilogues 0-2 go through a wave-private LDS scratch (behind the two stages) so that HBM sees 16-byte-per-lane stores of
// whole channel runs: the direct form (one 2-byte store per accumulator register) was store-ISSUE bound -- 64-128 store
// instructions per wave and tile cost 7-11 us of a 20-60 us tile (measured with the stores skipped).
template <int WMv, int EPI, bool RT>
__global__ __launch_bounds__(512, 2) void pconv_kernel(const PcParams P) {
  using C = PcCfg<WMv>;
  constexpr int NPW = RT ? C::WK : C::NP;            // DMA pieces a wave issues per stage (routed: the weights only)
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int r = lane & 31, h = lane >> 5;

  // ---- this workgroup's stream of tiles: the slabs of one spatial tile run on the same XCD (blocks b and b + 8 share one)
  const int bid = blockIdx.x;
  const int slab = (bid >> 3) % P.nslabs;
  const int stream = ((bid >> 3) / P.nslabs) * 8 + (bid & 7);
  const int nstreams = gridDim.x / P.nslabs;
  const int my_tiles = stream < P.ntiles ? (P.ntiles - stream + nstreams - 1) / nstreams : 0;
  if (my_tiles == 0) return;                         // uniform over the workgroup
  const int n_slab0 = slab * C::NSLAB;
  PcTile cur, nxt, dlt;
  cur.decode(stream, P.tiles_y, P.tiles_x);
  dlt.decode(nstreams, P.tiles_y, P.tiles_x);
  nxt = cur;

  // ---- fragment addresses (bytes inside a patch buffer): one per tap; the 4 row tiles of a wave are immediates
  uint32_t aaddr[9];
  {
    const int prow = wm * 4 + ((r & 3) >> 1), pcol = 2 * (r >> 2) + (r & 1);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int p = (prow + t / 3) * PC_RS + pcol + t % 3;
      aaddr[t] = (uint32_t)(p * 32 + 16 * (h ^ ((p >> 3) & 1)));
    }
  }
  const uint32_t baddr = (uint32_t)(2 * C::PATCH_BYTES + wn * 2048 + lane * 16);

  // ---- DMA source offsets of this wave's patch pieces i = wave + 8k (bytes from the tile's first pixel of the slice)
  uint32_t pvoff[C::PK];
  if (!RT) {
#pragma unroll
    for (int k = 0; k < C::PK; ++k) {
      const int i = wave + 8 * k;
      const int o = i * 1024 + lane * 16;
      const int p = o >> 5, slot = (o >> 4) & 1;
      const int row = p / PC_RS, xx = p - row * PC_RS;
      const int chunk = slot ^ ((p >> 3) & 1);
      pvoff[k] = (i < C::PATCH_INSTR && row < C::PROWS) ? (uint32_t)((row * P.W + xx) * 32 + chunk * 16) : BUF_OOB;
    }
  }
  const pc_rsrc_t wrs = pc_rsrc(P.wimg);
  const uint32_t wlane = (uint32_t)lane * 16u;

  float bias_v[2] = {0.f, 0.f};
  if (EPI < 2) {
#pragma unroll
    for (int j = 0; j < 2; ++j) bias_v[j] = P.bias[n_slab0 + wn * 64 + 32 * j + r];
  }

  // The next stage's pieces are issued one at a time BEHIND the MFMAs of each tap (piece n of this wave after tap n; the
  // patch pieces, which may miss L2, first): an LDS-DMA instruction holds its wave for a few hundred cycles, and issued in
  // a burst at the start of a stage -- by all waves, or by one wave of each SIMD -- that was 0.8-2 us of every 3-5 us stage
  // with the matrix pipes waiting (measured with the DMA skipped); spread out, the partner wave's MFMAs cover each one.
  pc_rsrc_t ns_rs = wrs;      // the next stage: patch source, weight offset, LDS buffer, "there is one"
  uint32_t ns_wsrc = 0;
  int ns_buf = 0;
  bool ns_on = false;
  auto next_stage = [&](const PcTile& tp, int slice, int buf, bool on) {
    if (!RT) {
      const char* base = P.x + ((((int64_t)tp.img * P.nslices + slice) * P.H + tp.ty * C::TY) * P.W + tp.tx * PC_TX) * 32;
      const int64_t left = P.x_end - base;
      ns_rs = pc_rsrc(base, left > 0xffff0000LL ? 0xffff0000u : (uint32_t)left);
    }
    ns_wsrc = (uint32_t)((slab * P.nslices + slice) * C::W_BYTES);
    ns_buf = buf;
    ns_on = on;
  };
  auto issue_piece = [&](int n) {        // n is a compile-time constant at every call
    if (!ns_on) return;
    if (!RT && n < C::PK) {
      const int i = wave + 8 * n;
      if (i < C::PATCH_INSTR)
        lds_dma16(ns_rs, smem + ns_buf * C::PATCH_BYTES + i * 1024, pvoff[n < C::PK ? n : 0], 0u);
    } else if (n < C::NP) {
      const int i = wave + 8 * (RT ? n : n - C::PK);
      if (i < C::W_INSTR && (!RT || n < C::WK))
        lds_dma16(wrs, smem + 2 * C::PATCH_BYTES + ns_buf * C::W_BYTES + i * 1024, wlane, ns_wsrc + i * 1024);
    }
  };

  // ---- routed patches (backward-data): the patch of a slice is BUILT in LDS from the block's pooled gradient and arg-max
  // bytes -- pre-pool gradient dY(y, x) = dP(y/2, x/2) where the stored byte equals the pixel's place in its window, zero
  // elsewhere (border of two pixels, rows / columns the pool dropped, dead windows) -- instead of being copied from a
  // materialised map four times the size.  A task = (pooled pixel of the (PROWS/2) x 17 under the patch, 8-channel half):
  // 16 bytes of dP + 8 arg-max bytes -> four 16-byte chunks (pc_route8), one per pixel of the window.  Thread t takes tasks
  // t, t + 512; the loads for stage q + 2 are issued during stage q (behind tap 1), routed into the other buffer during
  // stage q + 1 (behind tap 0), read by the MFMAs of stage q + 2.
  constexpr int RT_PR = C::PROWS / 2, RT_TASKS = RT_PR * 17 * 2, RT_K = (RT_TASKS + 511) / 512;
  bool rt_ok[RT_K];
  int rt_prow[RT_K], rt_pcol[RT_K];
  uint32_t rt_a0[RT_K], rt_a1[RT_K];
  const int rt_hb = threadIdx.x & 1;
  float4 rt_d[RT_K];
  uint2 rt_a[RT_K];
  PcTile rt_tile = cur;
  int rt_slice = 0, rt_count = 0;
  const int rt_total = my_tiles * P.nslices;
  if (RT) {
#pragma unroll
    for (int k = 0; k < RT_K; ++k) {
      const int tid = k * 512 + (int)threadIdx.x, pp = tid >> 1;
      rt_ok[k] = tid < RT_TASKS;
      rt_prow[k] = pp / 17;
      rt_pcol[k] = pp - rt_prow[k] * 17;
      const int p = 2 * rt_prow[k] * PC_RS + 2 * rt_pcol[k], sw = (p >> 3) & 1;     // p is even: p, p + 1 share the swizzle bit;
      rt_a0[k] = (uint32_t)(p * 32 + 16 * (rt_hb ^ sw));                            // the row below (p + 40) has the other one
      rt_a1[k] = (uint32_t)((p + PC_RS) * 32 + 16 * (rt_hb ^ sw ^ 1));
    }
  }
  auto rt_load = [&]() {           // ALWAYS issues its 2 RT_K loads (the counted waits rely on it); past the stream: zeros
    const bool on = rt_count < rt_total;
    const int64_t plane = ((int64_t)rt_tile.img * P.nslices + rt_slice) * P.Hq * P.Wq;
    const uint32_t pbytes = (uint32_t)(P.Hq * P.Wq * 32);
    const __amdgpu_buffer_rsrc_t rd = buf_rsrc(P.x + plane * 32, pbytes), ra = buf_rsrc(P.am_in + plane * 16, pbytes >> 1);
    const int py0 = rt_tile.ty * (C::TY / 2) - 1, px0 = rt_tile.tx * (PC_TX / 2) - 1;
#pragma unroll
    for (int k = 0; k < RT_K; ++k) {
      const int py = py0 + rt_prow[k], px = px0 + rt_pcol[k];
      const bool ok = on && rt_ok[k] && (unsigned)py < (unsigned)P.Hq && (unsigned)px < (unsigned)P.Wq;
      const uint32_t o = (uint32_t)((py * P.Wq + px) * 2 + rt_hb);
      rt_d[k] = buf_load16(rd, ok ? o * 16u : BUF_OOB);
      rt_a[k] = buf_load8(ra, ok ? o * 8u : BUF_OOB);
    }
    ++rt_count;
    if (++rt_slice == P.nslices) { rt_slice = 0; rt_tile.advance(dlt, P.tiles_y, P.tiles_x); }
  };
  auto rt_route = [&](int buf) {   // the registers -> the four pixels of each task's window, in patch buffer `buf`
    char* const dst = smem + buf * C::PATCH_BYTES;
#pragma unroll
    for (int k = 0; k < RT_K; ++k) {
      if (rt_ok[k]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<float4*>(dst + ((j >> 1) ? rt_a1[k] : rt_a0[k]) + (j & 1) * 32) = pc_route8(rt_d[k], rt_a[k], (uint32_t)j);
      }
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  char* const scr = smem + C::LDS + wave * C::SCR;     // wave-private epilogue scratch

  next_stage(nxt, 0, 0, true);
#pragma unroll
  for (int n = 0; n < NPW; ++n) issue_piece(n);
  if (RT) {
    rt_load();
    rt_route(0);
    rt_load();
  }
  int buf = 0;
  bool after_epilogue = false;
  for (int k = 0; k < my_tiles; ++k) {
    for (int slice = 0; slice < P.nslices; ++slice) {
      // this wave's pieces of the current stage have landed (they were issued BEFORE the epilogue's stores, if any)
      if (after_epilogue) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::template nstores<EPI>()) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      after_epilogue = false;
      __builtin_amdgcn_s_barrier();                       // everybody's have; everybody is done reading the other buffer
      int nsl = slice + 1;
      bool more = !PC_DBG(2);
      if (nsl == P.nslices) {
        nsl = 0;
        nxt.advance(dlt, P.tiles_y, P.tiles_x);
        more = more && k + 1 < my_tiles;
      }
      next_stage(nxt, nsl, buf ^ 1, more);
      const char* const pa = smem + buf * C::PATCH_BYTES;
      const char* const pb = smem + buf * C::W_BYTES + baddr;
      bf16x8 a[2][4], b[2][2];
      auto fetch = [&](int t, int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          a[set][i] = *reinterpret_cast<const bf16x8*>(pa + aaddr[t] + (i >> 1) * (2 * PC_RS * 32) + (i & 1) * 512);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          b[set][j] = *reinterpret_cast<const bf16x8*>(pb + t * (C::NT * 1024) + j * 1024);
      };
      fetch(0, 0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t < 8) fetch(t + 1, (t + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (!PC_DBG(4)) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = RT ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[t & 1][j], a[t & 1][i], acc[i][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t & 1][i], b[t & 1][j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(a[t & 1][i]));
#pragma unroll
          for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(b[t & 1][j]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (RT) {
          // routed: the patch of the next stage behind tap 0 (its registers were loaded a stage ago), the loads for the stage
          // after that behind tap 1, the weight pieces behind taps 2 ..
          if (t == 0) { if (ns_on) rt_route(buf ^ 1); }
          else if (t == 1) rt_load();
          else if (t < 8) issue_piece(t - 2);
        } else if (t == 0) {
          issue_piece(0);
          if (C::NP == 9) issue_piece(1);
        } else if (t < 8) {
          issue_piece(t + (C::NP == 9 ? 1 : 0));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      buf ^= 1;
    }
    // ---- epilogue of tile k, then clear the accumulators
    const int y0 = cur.ty * C::TY, x0 = cur.tx * PC_TX, img = cur.img;
    cur.advance(dlt, P.tiles_y, P.tiles_x);
    if (PC_DBG(1)) {
      if (acc[0][0][0] == 123.456f) static_cast<float*>(P.out)[0] = 1.f;     // keeps the accumulators alive
    } else if (EPI < 2) {
      // rounds of (pooled row ip of the wave, 32-channel tile j): 16 windows x 32 channels through the scratch.
      // EPI 0 (bf16, feeds the next block's patch DMA): C16 output, scratch [16-channel block][window][16]; EPI 1 (fp32, feeds
      // the L2 norm): NHWC, scratch [window][32]
      constexpr int ES = EPI == 0 ? 2 : 4;
      constexpr int AO = 16 * 32 * ES;                    // arg-max bytes behind the pooled values
      const int64_t plane = (int64_t)P.Hp * P.Wp;
#pragma unroll
      for (int ip = 0; ip < 2; ++ip) {
        const int py = (y0 >> 1) + wm * 2 + ip;
        const int px0 = x0 >> 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * ip + ii;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float best = acc[i][j][4 * g];
              int am = 0;
              if (acc[i][j][4 * g + 1] > best) { best = acc[i][j][4 * g + 1]; am = 1; }
              if (acc[i][j][4 * g + 2] > best) { best = acc[i][j][4 * g + 2]; am = 2; }
              if (acc[i][j][4 * g + 3] > best) { best = acc[i][j][4 * g + 3]; am = 3; }
              best += bias_v[j];
              const int wx = 8 * ii + 2 * g + h;
              if (EPI == 0) *reinterpret_cast<uint16_t*>(scr + (((r >> 4) * 16 + wx) * 16 + (r & 15)) * 2) = bf16_bits(best > 0.f ? best : 0.f);
              else *reinterpret_cast<float*>(scr + (wx * 32 + r) * 4) = best > 0.f ? best : 0.f;
              *reinterpret_cast<uint8_t*>(scr + AO + ((r >> 4) * 16 + wx) * 16 + (r & 15)) = best > 0.f ? (uint8_t)am : (uint8_t)4;
            }
          }
          asm volatile("" ::: "memory");       // scratch: the wave's LDS accesses execute in order; keep the compiler's order too
          {   // the stores are ALWAYS issued (rows below the map as out-of-range lanes): the next stage counts them
            const bool rowok = py < P.Hp;
            const int colt = n_slab0 + wn * 64 + 32 * j;
            const int64_t o_nhwc = (((int64_t)img * P.Hp + py) * P.Wp + px0) * P.N + colt;
            if (EPI == 0) {
              const int64_t o16 = (((int64_t)img * (P.N / 16) + colt / 16) * plane + (int64_t)py * P.Wp + px0) * 16;
              const __amdgpu_buffer_rsrc_t rp = buf_rsrc(static_cast<uint16_t*>(P.out) + o16);
              const int byte = lane * 16;                         // 1 KiB: lanes 0-31 block 0, lanes 32-63 block 1
              const int blk = byte >> 9, wx = (byte & 511) >> 5, inrun = byte & 31;
              const float4 v = *reinterpret_cast<const float4*>(scr + byte);
              const bool ok = rowok && px0 + wx < P.Wp;
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rp,
                                                     ok ? (int)(blk * (int)plane * 32 + wx * 32 + inrun) : (int)BUF_OOB, 0, 0);
            } else {
              const __amdgpu_buffer_rsrc_t rp = buf_rsrc(static_cast<float*>(P.out) + o_nhwc);
#pragma unroll
              for (int q = 0; q < 2; ++q) {                       // 1 KiB of pooled values per instruction
                const int byte = q * 1024 + lane * 16;
                const int wx = byte >> 7, inrun = byte & 127;
                const float4 v = *reinterpret_cast<const float4*>(scr + byte);
                const bool ok = rowok && px0 + wx < P.Wp;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rp,
                                                       ok ? (int)(wx * P.N * 4 + inrun) : (int)BUF_OOB, 0, 0);
              }
            }
            {   // 512 arg-max bytes, C16 like the activations: the lower 32 lanes, lane = (block, window)
              const int64_t a16 = (((int64_t)img * (P.N / 16) + colt / 16) * plane + (int64_t)py * P.Wp + px0) * 16;
              const __amdgpu_buffer_rsrc_t ra = buf_rsrc(P.amax + a16);
              const int blk = (lane >> 4) & 1, wx = lane & 15;
              const float4 v = *reinterpret_cast<const float4*>(scr + AO + (lane & 31) * 16);
              const bool ok = rowok && lane < 32 && px0 + wx < P.Wp;
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ra, ok ? (int)(blk * (int)plane * 16 + wx * 16) : (int)BUF_OOB, 0, 0);
            }
          }
          asm volatile("" ::: "memory");
        }
      }
    } else {
      // ---- backward-data epilogues.  The routed kernels multiply TRANSPOSED (weights as the A operand): lane (r, h) holds ONE
      // pixel of the row tile -- (dy, col) = ((r >> 1) & 1, 2 (r >> 2) + (r & 1)) -- and, per register group e >> 2, four CONSECUTIVE
      // channels 32 j + 8 (e >> 2) + 4 h ..: a group is one v_cvt_pk pair and one 8-byte LDS write (the pixel-in-registers form
      // needed 128 conversions and 128 two-byte writes per wave and row-tile round; the epilogue was 0.3-0.5 ms of a 3 ms launch)
      const int pdy = (r >> 1) & 1, pcol = 2 * (r >> 2) + (r & 1), pp_l = pdy * 16 + pcol;
      if (EPI == 2) {
        // [pixel][64 ch] bf16 through the scratch, pixel stride 144 bytes (two-way conflicts on the write, aligned 16-byte reads)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
              *reinterpret_cast<uint2*>(scr + pp_l * 144 + (32 * j + 8 * q4 + 4 * h) * 2) =
                  make_uint2(pack_bf16x2(acc[i][j][4 * q4], acc[i][j][4 * q4 + 1]), pack_bf16x2(acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]));
          asm volatile("" ::: "memory");
          const int ya = y0 + wm * 4 + 2 * (i >> 1), xa = x0 + 16 * (i & 1);
          const int64_t o0 = (((int64_t)img * P.Hc + ya) * P.Wc + xa) * P.N + n_slab0 + wn * 64;
          const __amdgpu_buffer_rsrc_t ro = buf_rsrc(static_cast<uint16_t*>(P.out) + o0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int pp = q * 8 + (lane >> 3), inrun = (lane & 7) * 16;
            const int dy = pp >> 4, col = pp & 15;
            const float4 v = *reinterpret_cast<const float4*>(scr + pp * 144 + inrun);
            const bool ok = ya + dy < P.Hc && xa + col < P.Wc;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro,
                                                   ok ? (int)((dy * P.Wc + col) * P.N * 2 + inrun) : (int)BUF_OOB, 0, 0);
          }
          asm volatile("" ::: "memory");
        }
      } else if (EPI == 4) {
        // C16 output [B][N/16][Hc][Wc][16] (the pooled gradient of the block below, read by its routed patches): per 32-channel
        // half j the scratch holds [2 blocks][32 pixels][48 bytes] (32 used); a 1-KiB store instruction = one block's 2 x 16 pixels
        const int64_t cplane = (int64_t)P.Hc * P.Wc;
        const int st_dy = lane >> 5, st_col = (lane >> 1) & 15;
        const int st_off = (st_dy * P.Wc + st_col) * 32 + (lane & 1) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ya = y0 + wm * 4 + 2 * (i >> 1), xa = x0 + 16 * (i & 1);
          const int64_t o0 = ((((int64_t)img * (P.N / 16) + (n_slab0 + wn * 64) / 16) * P.Hc + ya) * P.Wc + xa) * 16;
          const __amdgpu_buffer_rsrc_t ro = buf_rsrc(static_cast<uint16_t*>(P.out) + o0);
          const bool ok = ya + st_dy < P.Hc && xa + st_col < P.Wc;
          const int vo = ok ? st_off : (int)BUF_OOB;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
              *reinterpret_cast<uint2*>(scr + ((q4 >> 1) * 32 + pp_l) * 48 + (8 * (q4 & 1) + 4 * h) * 2) =
                  make_uint2(pack_bf16x2(acc[i][j][4 * q4], acc[i][j][4 * q4 + 1]), pack_bf16x2(acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]));
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < 2; ++u) {            // block 2 j + u: lane = (pixel, 16-byte half)
              const float4 v = *reinterpret_cast<const float4*>(scr + (u * 32 + (lane >> 1)) * 48 + (lane & 1) * 16);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, vo, (2 * j + u) * (int)cplane * 32, 0);
            }
            asm volatile("" ::: "memory");
          }
        }
      } else {
        // fp32 NHWC (tests): one dword per register, lane = pixel
        float* const out32 = static_cast<float*>(P.out);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ya = y0 + wm * 4 + 2 * (i >> 1) + pdy, xa = x0 + 16 * (i & 1) + pcol;
          const int64_t o0 = (int64_t)img * P.Hc * P.Wc * P.N + n_slab0 + wn * 64;
          const __amdgpu_buffer_rsrc_t ro = buf_rsrc(out32 + o0);
          const bool ok = ya < P.Hc && xa < P.Wc;
          const uint32_t vl = ok ? (uint32_t)(((ya * P.Wc + xa) * P.N + 4 * h) * 4) : BUF_OOB;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
              buf_store4(ro, acc[i][j][e], vl, (uint32_t)((32 * j + 8 * (e >> 2) + (e & 3)) * 4));
        }
      }
    }
    after_epilogue = !PC_DBG(1) && EPI != 3;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  }
}

// ------------------------------------------------------------------ packed weights
// image[slab][slice][tap][ntile][lane = (c, h)][e] (bf16), NT = nslab / 32 tiles per slab:
//   forward  (flip = 0): w[co = slab*nslab + 32*ntile + c][ci = 16*slice + 8h + e][ky][kx],          tap = 3*ky + kx
//   backward (flip = 1): w[co = 16*slice + 8h + e][ci = slab*nslab + 32*ntile + c][2 - ky][2 - kx]   (dX = dY_pad * flipped w)
__global__ void pconv_pack_kernel(const float* w, uint16_t* img, int Co, int Ci, int nslab, int flip) {
  const int Kc = flip ? Co : Ci, Nc = flip ? Ci : Co;       // reduction-side / output-side channel counts
  const int NT = nslab / 32, nslices = Kc / 16;
  const int64_t total = (int64_t)(Nc / nslab) * nslices * 9 * NT * 64;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ln = (int)(idx & 63);
  int64_t t = idx >> 6;
  const int ntile = (int)(t % NT); t /= NT;
  const int tap = (int)(t % 9); t /= 9;
  const int slice = (int)(t % nslices);
  const int slab = (int)(t / nslices);
  const int c = ln & 31, hh = ln >> 5;
  const int n = slab * nslab + 32 * ntile + c;
  const int ky = tap / 3, kx = tap - 3 * ky;
  uint32_t out[4];
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    float v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int kc = 16 * slice + 8 * hh + e + u;
      v[u] = flip ? w[((int64_t)kc * Ci + n) * 9 + (2 - ky) * 3 + (2 - kx)] : w[((int64_t)n * Ci + kc) * 9 + tap];
    }
    out[e >> 1] = pack_bf16x2(v[0], v[1]);
  }
  *reinterpret_cast<uint4*>(img + idx * 8) = make_uint4(out[0], out[1], out[2], out[3]);
}

// ------------------------------------------------------------------ weight gradient
// dW[(tap, ci), co] = sum over pixels of x[pixel + tap][ci] * dY[pixel][co]: the reduction runs over pixels, so both MFMA
// operands are reduction-major and reach the matrix cores through ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane
// group, transposed on the way).  A workgroup owns a WHOLE [9 taps x 64 ci] x [128 co] block of dW: wave (wci, wco) keeps the
// nine 32 x 32 accumulators (one per tap) of input-channel block wci and output-channel block wco -- 144 VGPRs -- and the
// workgroup streams 4 x 32-pixel tiles: per 16-pixel k-step a wave reads ONE dY fragment and nine x fragments (the taps are
// nine pixel shifts of the same LDS patch) for nine MFMAs.  Both operands are C16 in HBM and in LDS: a stage holds, per
// 16-channel block, the x patch 6 x 34 pixels x 32 bytes (four blocks) and the dY tile 4 x 32 pixels x 32 bytes (eight
// blocks); every DMA piece is one contiguous KiB of HBM.  Block strides are 128 (mod 256) bytes: the two blocks a
// transpose-read touches (channels 0-15 / 16-31 of the wave's 32) then sit on different banks, and its four pixels are 128
// contiguous bytes -- conflict-free.  Two stages in LDS, filled one stage ahead by the four waves of one half while the
// other half computes.  Tiles may overhang the map: dy_pad is zero there (vqa_pconv_dy_dims), so no masks.
constexpr int PW_XRS = 34;                                   // x patch row stride in pixels
constexpr int PW_XPC = 7;                                    // 1-KiB pieces per block of the patch (6 * 34 = 204 pixels of 32 bytes)
constexpr int PW_XB = PW_XPC * 1024 + 128;                   // block stride of the patch
constexpr int PW_X = 4 * PW_XB;
constexpr int PW_DB = 4 * 1024 + 128;                        // block stride of the dY tile (128 pixels of 32 bytes)
constexpr int PW_DY = 8 * PW_DB;
constexpr int PW_STAGE = PW_X + PW_DY;
constexpr int PW_LDS = 2 * PW_STAGE;

struct PwParams {
  const char* x; const char* x_end;       // C16 [B][Ci/16][H][W][16] bf16
  const char* dp;                          // pooled gradient, C16 [B][Co/16][Hp][Wp][16] bf16
  const char* am;                          // arg-max bytes, C16 [B][Co/16][Hp][Wp][16]
  float* slabs;                            // [grid][9][64][128]
  float* bslabs;                           // [grid][128]: bias-gradient partial sums (roles with ci0 == 0)
  int B, H, W, Ci, Co, Hp, Wp;
  int tiles_y, tiles_x, ntiles;            // 4 x 32-pixel tiles over the pool-covered map, per image / in all
  int roles_co, nroles;                    // Co / 128, (Ci / 64) * (Co / 128)
  int dbg;                                 // timing experiments only (VQA_PCONV_DBG): 2 = no DMA after the first stage, 4 = no MFMA
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

__global__ __launch_bounds__(512, 2) void pconv_wgrad_kernel(const PwParams P) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wci = wave >> 2, wco = wave & 3;
  const int bid = blockIdx.x;
  const int role = (bid >> 3) % P.nroles;
  const int stream = ((bid >> 3) / P.nroles) * 8 + (bid & 7);
  const int nstreams = gridDim.x / P.nroles;
  const int ci0 = (role / P.roles_co) * 64, co0 = (role % P.roles_co) * 128;
  const int my_tiles = stream < P.ntiles ? (P.ntiles - stream + nstreams - 1) / nstreams : 0;

  // ---- fragment addresses: lane = (h: pixels 8h.., grp: channels 16 grp.. = which C16 block, q: pixel in a 4-block, p4: 4-channel piece)
  const int h = lane >> 5, grp = (lane >> 4) & 1, q = (lane >> 2) & 3, p4 = lane & 3;
  const uint32_t a_lane = (uint32_t)((2 * wci + grp) * PW_XB + (8 * h + q) * 32 + p4 * 8);
  const uint32_t b_lane = (uint32_t)(PW_X + (2 * wco + grp) * PW_DB + (8 * h + q) * 32 + p4 * 8);

  // ---- x patch by DMA: the four waves of one half issue a stage's patch (the halves take turns, as in pconv_kernel): wave w4
  // fetches block w4 (7 pieces, gathered: 34 of a row's pixels).  The dY tile is ROUTED: thread (wave = block of 16 channels,
  // lane = (pooled row, pooled column, 8-channel half)) loads 16 bytes of dP + 8 arg-max bytes of one of the 2 x 16 windows
  // under the tile and writes the window's four pixels (pc_route8) -- the pre-pool gradient never exists in HBM.  Loads for
  // stage q + 2 are issued during stage q, routed into the other buffer at the start of stage q + 1.
  const int half = wave >> 2, w4 = wave & 3;
  uint32_t xvoff[PW_XPC];
#pragma unroll
  for (int i = 0; i < PW_XPC; ++i) {
    const int px = i * 32 + (lane >> 1);
    const int row = px / PW_XRS, xx = px - row * PW_XRS;
    xvoff[i] = row < 6 ? (uint32_t)((row * P.W + xx) * 32 + (lane & 1) * 16) : BUF_OOB;
  }
  const int rt_prow = lane >> 5, rt_pcol = (lane >> 1) & 15, rt_hb = lane & 1;
  const uint32_t rt_dst = (uint32_t)(PW_X + wave * PW_DB + (2 * rt_prow * 32 + 2 * rt_pcol) * 32 + rt_hb * 16);
  float4 rt_d;
  uint2 rt_a;

  PcTile nxt, dlt, rtt;
  nxt.decode(stream < P.ntiles ? stream : 0, P.tiles_y, P.tiles_x);
  dlt.decode(nstreams, P.tiles_y, P.tiles_x);
  rtt = nxt;
  int rt_count = 0;
  auto issue = [&](const PcTile& tp, int buf) {
    const int y0 = tp.ty * 4, x0 = tp.tx * 32;
    const char* xb = P.x + ((((int64_t)tp.img * (P.Ci / 16) + ci0 / 16 + w4) * P.H + y0) * P.W + x0) * 32;
    const int64_t xl = P.x_end - xb;
    const pc_rsrc_t rx = pc_rsrc(xb, xl > 0xffff0000LL ? 0xffff0000u : (uint32_t)xl);
    char* const st = smem + buf * PW_STAGE;
#pragma unroll
    for (int i = 0; i < PW_XPC; ++i)
      lds_dma16(rx, st + w4 * PW_XB + i * 1024, xvoff[i], 0u);
  };
  auto rt_load = [&]() {         // ALWAYS two loads (the counted wait of the next stage relies on it); past the stream: zeros
    const bool on = rt_count < my_tiles;
    const int64_t plane = ((int64_t)rtt.img * (P.Co / 16) + co0 / 16 + wave) * P.Hp * P.Wp;
    const uint32_t pbytes = (uint32_t)(P.Hp * P.Wp * 32);
    const __amdgpu_buffer_rsrc_t rd = buf_rsrc(P.dp + plane * 32, pbytes), ra = buf_rsrc(P.am + plane * 16, pbytes >> 1);
    const int py = rtt.ty * 2 + rt_prow, px = rtt.tx * 16 + rt_pcol;
    const bool ok = on && py < P.Hp && px < P.Wp;
    const uint32_t o = (uint32_t)((py * P.Wp + px) * 2 + rt_hb);
    rt_d = buf_load16(rd, ok ? o * 16u : BUF_OOB);
    rt_a = buf_load8(ra, ok ? o * 8u : BUF_OOB);
    ++rt_count;
    rtt.advance(dlt, P.tiles_y, P.tiles_x);
  };
  // bias gradient = sum of the pooled gradient over the live windows (arg-max code != 4): every (window, 8 channels) piece of dP
  // passes through exactly one thread of every role here, so the roles with ci0 == 0 sum it on the way (8 fp32 per thread, combined
  // over the wave's 32 windows at the end) -- the separate pass over dP + arg-max (0.6 ms per step at 448 x 448) is gone
  float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool want_bias = ci0 == 0;
  auto rt_route = [&](int buf) {
    char* const dst = smem + buf * PW_STAGE + rt_dst;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(dst + ((j >> 1) * 32 + (j & 1)) * 32) = pc_route8(rt_d, rt_a, (uint32_t)j);
    if (want_bias) {
      const uint32_t dd[4] = {__float_as_uint(rt_d.x), __float_as_uint(rt_d.y), __float_as_uint(rt_d.z), __float_as_uint(rt_d.w)};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t code = ((k < 4 ? rt_a.x : rt_a.y) >> (8 * (k & 3))) & 0xffu;
        const uint32_t bits = (k & 1) ? (dd[k >> 1] & 0xffff0000u) : (dd[k >> 1] << 16);
        bs[k] += __uint_as_float(bits & ((code >> 2) - 1u));           // codes 0..3 keep, 4 (dead window) clears
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  auto trread = [](const char* ptr) -> s16x4 { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(ptr)); };
  auto frag = [&](const char* ptr) -> bf16x8 {          // pixels q .. and q + 4 ..: 4 pixels are 128 bytes
    const s16x4 lo = trread(ptr), hi = trread(ptr + 128);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  if (my_tiles > 0) {
    if (half == 1) issue(nxt, 0);
    rt_load();
    rt_route(0);
    rt_load();
  }
  int buf = 0;
  for (int k = 0; k < my_tiles; ++k) {
    // the x pieces of this stage have landed; the two routed loads issued after them (mid-stage) may still be in flight
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    nxt.advance(dlt, P.tiles_y, P.tiles_x);
    if (k + 1 < my_tiles && half == buf && !PC_DBG(2)) issue(nxt, buf ^ 1);
    const char* const xa = smem + buf * PW_STAGE + a_lane;
    const char* const da = smem + buf * PW_STAGE + b_lane;
    bf16x8 a[9], b[2];
    // k-step s: output row s >> 1 of the tile, pixels 16 (s & 1) .. + 15
    b[0] = frag(da);
#pragma unroll
    for (int t = 0; t < 9; ++t) a[t] = frag(xa + ((t / 3) * PW_XRS + t % 3) * 32);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < 7) b[(s + 1) & 1] = frag(da + (((s + 1) >> 1) * 32 + 16 * ((s + 1) & 1)) * 32);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (!PC_DBG(4)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[s & 1], acc[t], 0, 0, 0);
        else asm volatile("" ::"v"(a[t]), "v"(b[s & 1]));
        if (s < 7)
          a[t] = frag(xa + ((((s + 1) >> 1) + t / 3) * PW_XRS + 16 * ((s + 1) & 1) + t % 3) * 32);
      }
      // the next stage's dY tile: routed from the registers loaded a stage ago (hipcc waits for those two loads: by now nothing
      // older is in flight), then the loads for the stage after it.  The two waves of a SIMD do it at different k-steps, so the
      // VALU / ds_write burst of one runs under the other's MFMAs -- the half that issued this stage's DMA pieces later (hipcc's
      // wait for the two loads is a vmcnt(0): it would wait for pieces issued just before, too).
      if (s == (half == buf ? 5 : 2)) {
        __builtin_amdgcn_sched_barrier(0);
        if (k + 1 < my_tiles) rt_route(buf ^ 1);
        rt_load();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    buf ^= 1;
  }
  // ---- bias partial sums: lanes that share (block = wave, half) differ in lane bits 1..5
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int o = 2; o < 64; o <<= 1) bs[k] += __shfl_xor(bs[k], o, 64);
  }
  if (lane < 2) {
#pragma unroll
    for (int k = 0; k < 8; ++k) P.bslabs[(int64_t)bid * 128 + wave * 16 + lane * 8 + k] = bs[k];
  }
  // ---- this workgroup's fp32 slab [tap][64 ci][128 co] (zeros when it had no tile)
  float* const slab = P.slabs + (int64_t)bid * (9 * 64 * 128);
  const int c = lane & 31;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
      slab[(t * 64 + wci * 32 + row) * 128 + wco * 32 + c] = acc[t][e];
    }
}

// dw[co][ci][tap] = sum over the workgroups of role (ci / 64, co / 128) of slab[tap][ci % 64][co % 128], in workgroup order;
// the last Co threads: dbias[co] = sum over the workgroups of role (0, co / 128) of bslab[co % 128]
__global__ __launch_bounds__(256) void pconv_wgrad_reduce_kernel(const float* slabs, const float* bslabs, float* dw, float* dbias,
                                                                int Ci, int Co, int grid, int nroles, int roles_co) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // (tap, ci, co), co fastest: coalesced slab reads
  if (idx >= 9 * Ci * Co + Co) return;
  if (idx >= 9 * Ci * Co) {
    const int co = idx - 9 * Ci * Co, role = co / 128;
    float s = 0.f;
    for (int b = 0; b < grid; ++b)
      if ((b >> 3) % nroles == role) s += bslabs[(int64_t)b * 128 + (co & 127)];
    dbias[co] = s;
    return;
  }
  const int co = idx % Co, ci = (idx / Co) % Ci, tap = idx / (Co * Ci);
  const int role = (ci / 64) * roles_co + co / 128;
  const int off = (tap * 64 + (ci & 63)) * 128 + (co & 127);
  float s = 0.f;
  for (int b = 0; b < grid; ++b)
    if ((b >> 3) % nroles == role) s += slabs[(int64_t)b * (9 * 64 * 128) + off];
  dw[((int64_t)co * Ci + ci) * 9 + tap] = s;
}

// ------------------------------------------------------------------ host side
static int pc_nslab(int N) { return N % 128 == 0 ? 128 : 64; }

template <int WMv, int EPI, bool RT>
static int pc_launch(const PcParams& P, hipStream_t s) {
  using C = PcCfg<WMv>;
  auto kern = pconv_kernel<WMv, EPI, RT>;
  int rc = ensure_dyn_smem(reinterpret_cast<const void*>(kern), C::LDS_ALL, "attr(pconv)");
  if (rc) return rc;
  int grid = 256;                                   // one workgroup per CU (120 KiB of LDS each)
  grid -= grid % (8 * P.nslabs);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), C::LDS_ALL, s, P);
  return check_hip(hipGetLastError(), "pconv launch");
}

// epi: 0 / 1 forward (pooled bf16 C16 / fp32 NHWC) from a C16 map; 2 / 3 / 4 backward-data (dX bf16 NHWC / fp32 NHWC / bf16 C16)
// from the routed pooled gradient
static int pc_run(PcParams P, int epi, hipStream_t s) {
  const int nslab = pc_nslab(P.N);
  const int TY = nslab == 128 ? 16 : 32;
  P.nslabs = P.N / nslab;
  P.nslices = P.Cin / 16;
  P.tiles_y = (P.Hc + TY - 1) / TY;
  P.tiles_x = (P.Wc + PC_TX - 1) / PC_TX;
  P.ntiles = P.B * P.tiles_y * P.tiles_x;
  {
    const char* e = getenv("VQA_PCONV_DBG");
    P.dbg = e ? atoi(e) : 0;
  }
  if (nslab == 128) {
    switch (epi) {
      case 0: return pc_launch<4, 0, false>(P, s);
      case 1: return pc_launch<4, 1, false>(P, s);
      case 2: return pc_launch<4, 2, true>(P, s);
      case 3: return pc_launch<4, 3, true>(P, s);
      default: return pc_launch<4, 4, true>(P, s);
    }
  }
  switch (epi) {
    case 0: return pc_launch<8, 0, false>(P, s);
    case 1: return pc_launch<8, 1, false>(P, s);
    case 2: return pc_launch<8, 2, true>(P, s);
    case 3: return pc_launch<8, 3, true>(P, s);
    default: return pc_launch<8, 4, true>(P, s);
  }
}

}  // namespace vqa

using namespace vqa;

extern "C" {

int vqa_pconv_supported(int H, int W, int Cin, int Cout, int stride) {
  if (stride != 1 || H < 4 || W < 4 || Cin % 16 || Cout % 64 || Cin <= 0 || Cout <= 0) return 0;
  if (Cin > 4096 || Cout > 4096) return 0;
  // 32-bit byte offsets inside one tile's patch / one output row pair, and 8 * nslabs must divide the 256-workgroup grid
  if ((int64_t)40 * W * (Cin > Cout ? Cin : Cout) * 2 >= (1LL << 31)) return 0;
  const int ns_f = Cout / pc_nslab(Cout);
  if (256 % (8 * ns_f)) return 0;
  if (Cin % 64 == 0 && 256 % (8 * (Cin / pc_nslab(Cin)))) return 0;     // backward-data: the roles of Cin and Cout swap
  return 1;
}

int64_t vqa_pconv_weights_bytes(int Cin, int Cout) { return (int64_t)9 * Cin * Cout * 2; }

int vqa_pconv_pack_weights(const float* w, void* wf_img, void* wd_img, int Co, int Ci, vqa_stream_t stream) {
  VQA_REQUIRE(w && (wf_img || wd_img) && Co % 64 == 0 && Ci % 16 == 0, "vqa_pconv_pack_weights: bad args Co=%d Ci=%d", Co, Ci);
  hipStream_t s = (hipStream_t)stream;
  const int64_t chunks = (int64_t)9 * Ci * Co / 8;
  if (wf_img) {
    hipLaunchKernelGGL(pconv_pack_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, w,
                       static_cast<uint16_t*>(wf_img), Co, Ci, pc_nslab(Co), 0);
    int rc = check_hip(hipGetLastError(), "pconv_pack(fwd) launch");
    if (rc) return rc;
  }
  if (wd_img) {
    VQA_REQUIRE(Ci % 64 == 0 && Co % 16 == 0, "vqa_pconv_pack_weights: the backward image needs Ci %% 64 == 0 (Ci=%d)", Ci);
    hipLaunchKernelGGL(pconv_pack_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, w,
                       static_cast<uint16_t*>(wd_img), Co, Ci, pc_nslab(Ci), 1);
    return check_hip(hipGetLastError(), "pconv_pack(bwd) launch");
  }
  return VQA_OK;
}

int vqa_pconv_fwd(const void* x, const void* wf_img, const float* bias, void* pooled, int pooled_is_bf16, uint8_t* argmax,
                  int B, int H, int W, int Ci, int Co, int tag, vqa_stream_t stream) {
  VQA_REQUIRE(x && wf_img && bias && pooled && argmax && B > 0, "vqa_pconv_fwd: null pointer");
  VQA_REQUIRE(vqa_pconv_supported(H, W, Ci, Co, 1), "vqa_pconv_fwd: unsupported shape H=%d W=%d Ci=%d Co=%d", H, W, Ci, Co);
  PcParams P{};
  P.x = static_cast<const char*>(x);
  P.x_end = P.x + (int64_t)B * H * W * Ci * 2;
  P.wimg = static_cast<const char*>(wf_img);
  P.bias = bias; P.out = pooled; P.amax = argmax;
  P.B = B; P.H = H; P.W = W; P.Cin = Ci; P.N = Co;
  P.Hp = (H - 2) / 2; P.Wp = (W - 2) / 2;
  VQA_REQUIRE(P.Hp > 0 && P.Wp > 0, "vqa_pconv_fwd: image too small");
  P.Hc = 2 * P.Hp; P.Wc = 2 * P.Wp;
  hipStream_t s = (hipStream_t)stream;
  set_launch_tag(tag);
  ProfScope prof(VQA_K_CONV_FWD, s);
  return pc_run(P, pooled_is_bf16 ? 0 : 1, s);
}

int vqa_pconv_dgrad(const void* dpooled, const uint8_t* argmax, const void* wd_img, void* dx, int dx_mode, int B, int H, int W,
                    int Ci, int Co, int tag, vqa_stream_t stream) {
  VQA_REQUIRE(dpooled && argmax && wd_img && dx && B > 0, "vqa_pconv_dgrad: null pointer");
  VQA_REQUIRE(dx_mode >= 0 && dx_mode <= 2, "vqa_pconv_dgrad: dx_mode %d (0 fp32 NHWC, 1 bf16 NHWC, 2 bf16 C16)", dx_mode);
  VQA_REQUIRE(vqa_pconv_supported(H, W, Ci, Co, 1) && Ci % 64 == 0,
              "vqa_pconv_dgrad: unsupported shape H=%d W=%d Ci=%d Co=%d", H, W, Ci, Co);
  PcParams P{};
  P.x = static_cast<const char*>(dpooled);                  // C16 [B][Co/16][Hq][Wq][16]: the patches are routed from it
  P.am_in = reinterpret_cast<const char*>(argmax);
  P.Hq = (H - 2) / 2; P.Wq = (W - 2) / 2;
  VQA_REQUIRE(P.Hq > 0 && P.Wq > 0, "vqa_pconv_dgrad: image too small");
  P.x_end = P.x + (int64_t)B * P.Hq * P.Wq * Co * 2;
  P.wimg = static_cast<const char*>(wd_img);
  P.out = dx;
  P.B = B; P.H = H + 2; P.W = W + 2; P.Cin = Co; P.N = Ci;   // the (virtual) padded pre-pool gradient map
  P.Hc = H; P.Wc = W;
  hipStream_t s = (hipStream_t)stream;
  set_launch_tag(tag);
  ProfScope prof(VQA_K_CONV_DGRAD, s);
  return pc_run(P, dx_mode == 0 ? 3 : dx_mode == 1 ? 2 : 4, s);
}

static int pw_roles(int Ci, int Co) { return (Ci / 64) * (Co / 128); }
int vqa_pconv_wgrad_supported(int H, int W, int Ci, int Co) {
  if (H < 4 || W < 4 || Ci <= 0 || Co <= 0 || Ci % 64 || Co % 128) return 0;
  const int r = pw_roles(Ci, Co);
  if (r != 1 && r != 2 && r != 4 && r != 8) return 0;
  if ((int64_t)8 * (W + 34) * (Ci > Co ? Ci : Co) * 2 >= (1LL << 31)) return 0;
  return 1;
}

int64_t vqa_pconv_wgrad_workspace_bytes(int B, int H, int W, int Ci, int Co) {
  if (!vqa_pconv_wgrad_supported(H, W, Ci, Co) || B <= 0) return 0;
  return ((int64_t)256 * 9 * 64 * 128 + (int64_t)256 * 128) * 4;
}

int vqa_pconv_wgrad(const void* x, const void* dpooled, const uint8_t* argmax, float* dw, float* dbias, int B, int H, int W,
                    int Ci, int Co, float* workspace, int64_t workspace_bytes, int tag, vqa_stream_t stream) {
  VQA_REQUIRE(x && dpooled && argmax && dw && dbias && workspace && B > 0, "vqa_pconv_wgrad: null pointer");
  VQA_REQUIRE(vqa_pconv_wgrad_supported(H, W, Ci, Co), "vqa_pconv_wgrad: unsupported shape H=%d W=%d Ci=%d Co=%d", H, W, Ci, Co);
  VQA_REQUIRE(Co <= 2048, "vqa_pconv_wgrad: Co=%d above 2048", Co);
  const int64_t need = vqa_pconv_wgrad_workspace_bytes(B, H, W, Ci, Co);
  if (workspace_bytes < need) {
    set_error("vqa_pconv_wgrad: workspace %lld < %lld", (long long)workspace_bytes, (long long)need);
    return VQA_ERR_WORKSPACE;
  }
  const int Hp = (H - 2) / 2, Wp = (W - 2) / 2;
  VQA_REQUIRE(Hp > 0 && Wp > 0, "vqa_pconv_wgrad: image too small");
  PwParams P{};
  P.x = static_cast<const char*>(x); P.x_end = P.x + (int64_t)B * H * W * Ci * 2;
  P.dp = static_cast<const char*>(dpooled); P.am = reinterpret_cast<const char*>(argmax);
  P.slabs = workspace;
  P.bslabs = workspace + (int64_t)256 * 9 * 64 * 128;
  P.B = B; P.H = H; P.W = W; P.Ci = Ci; P.Co = Co; P.Hp = Hp; P.Wp = Wp;
  P.tiles_y = (2 * Hp + 3) / 4; P.tiles_x = (2 * Wp + 31) / 32;
  P.ntiles = B * P.tiles_y * P.tiles_x;
  P.roles_co = Co / 128; P.nroles = pw_roles(Ci, Co);
  {
    const char* e = getenv("VQA_PCONV_DBG");
    P.dbg = e ? atoi(e) : 0;
  }
  hipStream_t s = (hipStream_t)stream;
  set_launch_tag(tag);
  ProfScope prof(VQA_K_CONV_WGRAD, s);
  int rc = ensure_dyn_smem(reinterpret_cast<const void*>(pconv_wgrad_kernel), PW_LDS, "attr(pconv_wgrad)");
  if (rc) return rc;
  const int grid = 256;
  hipLaunchKernelGGL(pconv_wgrad_kernel, dim3(grid), dim3(512), PW_LDS, s, P);
  rc = check_hip(hipGetLastError(), "pconv_wgrad launch");
  if (rc) return rc;
  hipLaunchKernelGGL(pconv_wgrad_reduce_kernel, dim3((9 * Ci * Co + Co + 255) / 256), dim3(256), 0, s, workspace, P.bslabs, dw, dbias,
                     Ci, Co, grid, P.nroles, P.roles_co);
  return check_hip(hipGetLastError(), "pconv_wgrad_reduce launch");
}

}  // extern "C"
