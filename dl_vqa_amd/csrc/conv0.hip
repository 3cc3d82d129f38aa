// First conv block (Cin <= 3, stride 1) — dedicated kernels.
//
// Reference: models/model.py:80-82, first iteration (Conv2d(3, 64, k=3) + ReLU + MaxPool2d(2,2)).
//
// Why not the generic implicit GEMM (conv.hip): K = 9*Cin = 27 is tiny, so a 128 x 64 x 32 tile spends its
// time staging operands and pads K to 64 (2.4x the MFMA work); the layer is 4 % of the FLOPs but was
// 9 % of the step.  Here
//   forward : the input patch (6 image rows x all columns, PLANAR = the caller's NCHW, so no NHWC
//             conversion pass at all) sits in LDS, the 27 x Co weights sit in REGISTERS as MFMA B
//             fragments for the whole workgroup, and a wave streams 32-pixel M-tiles (8 pool windows):
//             14 ds_read_b32 + 28 MFMA per tile, then the same in-register bias/ReLU/pool/arg-max epilogue.
//   wgrad   : persistent workgroups walk pool-window rows; A = the 27 taps of a pixel read from the planar
//             LDS patch (lane = tap), B = dY routed from the pooled gradient (staged in LDS with 16-byte
//             loads); each wave keeps its 32 x Co partial dW in accumulators for its whole lifetime and
//             writes ONE slab at the end (deterministic two-level reduction, no atomics).
// LDS row strides are chosen per kernel so that the 32 lanes of a fragment read hit 32 different banks.
#include "x3_core.hpp"

namespace vqa {

constexpr int C0_MAX_NS = 18;   // k2-steps for Cin = 4

__host__ __device__ inline int c0_round_stride(int W, int want_mod) {
  int rs = W + 2;
  while (rs % 32 != want_mod) ++rs;
  return rs;
}

// The image may arrive as the dataset's fp16 features (preprocessing/preprocess_images.py:39-53 stores them as float16;
// the reference widens every sample on the host, data_preprocessing.py:167-176): xh != 0 reads __half NCHW rows and
// widens them where the patch is staged -- exact, so results are bit-identical to a widened fp32 copy, without the
// 0.15 GB read + 0.3 GB write + 0.3 GB re-read of a separate conversion pass per step.
__device__ __forceinline__ float4 c0_load4(const void* x, int64_t e, int xh) {
  if (xh) {
    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
    const h16x4 hv = *reinterpret_cast<const h16x4*>(static_cast<const uint16_t*>(x) + e);     // e % 4 == 0: 8-byte aligned
    return make_float4((float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]);
  }
  return *reinterpret_cast<const float4*>(static_cast<const float*>(x) + e);
}

// ------------------------------------------------------------------ forward
// grid (ceil(Hp/C0_FR), B); 256 threads; dynamic LDS = CI * C0_PR * RS floats (RS % 32 == 16).
// A workgroup covers C0_FR pool rows (2*C0_FR + 2 image rows): with 4 rows the patch staging, the barrier and
// the 28 weight-fragment loads are spread over 14 M-tiles per wave instead of 7.  The tile loop has no
// integer division (window row by compares, 24-bit multiplies) and stores through a buffer resource based at
// the workgroup's first window (lane part one VGPR per tile, group / column block in the scalar offset).
constexpr int C0_FR = 4;
constexpr int C0_PR = 2 * C0_FR + 2;

// OB: 0 = pooled is stored as fp32, 1 = as bf16 (the bf16 path's P_0), 2 = x3-packed (the fp32x3 path: vqa_x3_pack's form)
template <int CI, int TN, int OB>
__global__ __launch_bounds__(256, 4) void conv0_fwd_kernel(const void* __restrict__ x, int xh, const float* __restrict__ w,
                                                        const float* __restrict__ bias, void* pooled_, uint8_t* amax,
                                                        int H, int W, int Hp, int Wp, int RS) {
  float* pooled = static_cast<float*>(pooled_);
  uint16_t* pooled16 = static_cast<uint16_t*>(pooled_);
  extern __shared__ __attribute__((aligned(16))) float patch[];
  constexpr int K = 9 * CI, NS = (K + 1) / 2, Co = 32 * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int b = blockIdx.y, py0 = C0_FR * blockIdx.x;
  const int nwr = min(C0_FR, Hp - py0);
  const int y0 = 2 * py0, nrows = 2 * nwr + 2;
  const int plane = C0_PR * RS;
  // stage the planar patch: rows y0 .. y0+nrows-1 of every input channel; a wave takes whole rows
  for (int r = wave; r < CI * C0_PR; r += 4) {
    const int c = r / C0_PR, rr = r - c * C0_PR;        // wave-uniform
    const int64_t src = ((int64_t)(b * CI + c) * H + y0 + rr) * W;
    float* dst = patch + c * plane + rr * RS;
    for (int c4 = lane; c4 < W / 4; c4 += 64) {
      const float4 v = rr < nrows ? c0_load4(x, src + 4 * c4, xh) : f4zero();
      dst[4 * c4] = v.x; dst[4 * c4 + 1] = v.y; dst[4 * c4 + 2] = v.z; dst[4 * c4 + 3] = v.w;
    }
  }
  // weights as B fragments, tap offsets as per-lane constants
  float bf[NS][TN];
  int koff[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int k = 2 * s + h;
    const bool kok = k < K;
    const int kk = kok ? k : 0;
    const int c = kk / 9, t = kk - 9 * c, ky = t / 3, kx = t - 3 * ky;
    koff[s] = c * plane + ky * RS + kx;
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[s][j] = kok ? w[(int64_t)(32 * j + l31) * K + kk] : 0.f;
  }
  float bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bv[j] = bias[32 * j + l31];
  __syncthreads();

  const int nwin = nwr * Wp, ntiles = (nwin + 7) / 8;
  // windows of the workgroup are consecutive in memory: offset = first window + wo * Co
  const int64_t o0 = (int64_t)(b * Hp + py0) * Wp * Co;
  const __amdgpu_buffer_rsrc_t rp = OB == 2 ? buf_rsrc(pooled16 + 3 * o0) : OB == 1 ? buf_rsrc(pooled16 + o0) : buf_rsrc(pooled + o0),
                               ra = buf_rsrc(amax + o0);
  for (int t = wave; t < ntiles; t += 4) {
    // A rows: row i = 4*window + pixel (the engine's window-in-4-registers layout)
    int wdx = 8 * t + (l31 >> 2);
    if (wdx >= nwin) wdx = 0;
    const int wr = (wdx >= Wp ? 1 : 0) + (wdx >= 2 * Wp ? 1 : 0) + (wdx >= 3 * Wp ? 1 : 0);
    const int px = wdx - __mul24(wr, Wp), j4 = l31 & 3;
    const float* ap = patch + __mul24(2 * wr + (j4 >> 1), RS) + 2 * px + (j4 & 1);
    f32x16 acc[TN];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float a = ap[koff[s]];
#pragma unroll
      for (int j = 0; j < TN; ++j)   // the first MFMA takes the constant 0 as C: no 16 v_mov per accumulator tile
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bf[s][j], s == 0 ? zero : acc[j], 0, 0, 0);
    }
    const bool inner = 8 * t + 8 <= nwin;                       // uniform
    if (OB == 0) {
      // fp32 output (the headline path): the tile's 8 windows x Co channels go through a wave-private LDS scratch behind the
      // patch -- pooled [window][Co] fp32, arg-max [window][Co] bytes, both contiguous in NHWC -- and leave as 16-byte-per-lane
      // stores: 2 TN + 1 instead of 8 TN four-byte and 8 TN one-byte store instructions per tile (store-issue bound, as the
      // bf16 kernels were)
      char* const scr = reinterpret_cast<char*>(patch) + CI * C0_PR * RS * 4 + wave * (8 * Co * 5);
      char* const sam = scr + 8 * Co * 4;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float best = acc[j][4 * g];
          int a = 0;
          if (acc[j][4 * g + 1] > best) { best = acc[j][4 * g + 1]; a = 1; }
          if (acc[j][4 * g + 2] > best) { best = acc[j][4 * g + 2]; a = 2; }
          if (acc[j][4 * g + 3] > best) { best = acc[j][4 * g + 3]; a = 3; }
          best += bv[j];
          const int win = 2 * g + h;
          *reinterpret_cast<float*>(scr + (win * Co + 32 * j + l31) * 4) = best > 0.f ? best : 0.f;
          *reinterpret_cast<uint8_t*>(sam + win * Co + 32 * j + l31) = best > 0.f ? (uint8_t)a : (uint8_t)4;
        }
      }
      asm volatile("" ::: "memory");        // the wave's LDS accesses execute in order; keep the compiler's order as well
#pragma unroll
      for (int q = 0; q < TN; ++q) {         // pooled: 8 * Co * 4 bytes = TN KiB
        const int byte = q * 1024 + lane * 16;
        const int win = byte / (Co * 4);
        const float4 v = *reinterpret_cast<const float4*>(scr + byte);
        const bool ok = inner || 8 * t + win < nwin;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rp, ok ? (int)(8 * t * Co * 4 + byte) : (int)BUF_OOB, 0, 0);
      }
      {
        const int byte = lane * 16;
        if (byte < 8 * Co) {                 // arg-max: 8 windows x Co bytes
          const int win = byte / Co;
          const float4 v = *reinterpret_cast<const float4*>(sam + byte);
          const bool ok = inner || 8 * t + win < nwin;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ra, ok ? (int)(8 * t * Co + byte) : (int)BUF_OOB, 0, 0);
        }
      }
      asm volatile("" ::: "memory");
      continue;
    }
    const uint32_t vl = (uint32_t)__mul24(8 * t + h, Co) + (uint32_t)l31;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bool ok = inner || 8 * t + 2 * g + h < nwin;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float best = acc[j][4 * g];
        int a = 0;
        if (acc[j][4 * g + 1] > best) { best = acc[j][4 * g + 1]; a = 1; }
        if (acc[j][4 * g + 2] > best) { best = acc[j][4 * g + 2]; a = 2; }
        if (acc[j][4 * g + 3] > best) { best = acc[j][4 * g + 3]; a = 3; }
        best += bv[j];
        const uint32_t so = (uint32_t)(2 * g * Co + 32 * j);
        if (OB == 2) {
          uint16_t sh, sm, sl;
          split1(best > 0.f ? best : 0.f, sh, sm, sl);
          const uint32_t vx = ok ? 6u * (uint32_t)__mul24(8 * t + h, Co) + x3p_lane(l31) : BUF_OOB;
          buf_store2(rp, sh, vx, 6u * so);
          buf_store2(rp, sm, vx, 6u * so + 8u);
          buf_store2(rp, sl, vx, 6u * so + 16u);
        } else if (OB == 1) buf_store2(rp, bf16_bits(best > 0.f ? best : 0.f), ok ? 2u * vl : BUF_OOB, 2u * so);
        else buf_store4(rp, best > 0.f ? best : 0.f, ok ? 4u * vl : BUF_OOB, 4u * so);
        buf_store1(ra, best > 0.f ? (uint8_t)a : (uint8_t)4, ok ? vl : BUF_OOB, so);
      }
    }
  }
}

// ------------------------------------------------------------------ forward on bf16 MFMA (bf16 path, configs[3])
// Same patch staging (fp32, planar), but the 27 taps (padded to 32) are two v_mfma_f32_32x32x16_bf16 k-steps: a lane
// gathers its 2 x 8 taps from the patch, rounds them to bf16 (the image is rounded where it is consumed, as the weights
// are) and issues 2 x TN MFMAs per 32-pixel tile instead of 14 x TN fp32 ones -- the kernel becomes store-bound.
// C16: the pooled map is written channel-blocked, [B][Co/16][Hp][Wp][16] bf16 -- what the patch convolutions of the next
// block cut their LDS patches from (csrc/conv_patch_bf16.hip); the arg-max bytes stay NHWC.
template <int CI, int TN, bool C16 = false>
__global__ __launch_bounds__(256, 4) void conv0_fwd_bf16_kernel(const void* __restrict__ x, int xh, const float* __restrict__ w,
                                                             const float* __restrict__ bias, uint16_t* pooled16,
                                                             uint8_t* amax, int H, int W, int Hp, int Wp, int RS) {
  extern __shared__ __attribute__((aligned(16))) float patch[];
  constexpr int K = 9 * CI, Co = 32 * TN;
  static_assert(K <= 32, "two 16-deep k-steps cover at most 32 taps");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int b = blockIdx.y, py0 = C0_FR * blockIdx.x;
  const int nwr = min(C0_FR, Hp - py0);
  const int y0 = 2 * py0, nrows = 2 * nwr + 2;
  const int plane = C0_PR * RS;
  for (int r = wave; r < CI * C0_PR; r += 4) {
    const int c = r / C0_PR, rr = r - c * C0_PR;
    const int64_t src = ((int64_t)(b * CI + c) * H + y0 + rr) * W;
    float* dst = patch + c * plane + rr * RS;
    for (int c4 = lane; c4 < W / 4; c4 += 64) {
      const float4 v = rr < nrows ? c0_load4(x, src + 4 * c4, xh) : f4zero();
      dst[4 * c4] = v.x; dst[4 * c4 + 1] = v.y; dst[4 * c4 + 2] = v.z; dst[4 * c4 + 3] = v.w;
    }
  }
  // weights as bf16 B fragments (element e of k-step ks: tap k = 16 ks + 8 h + e), tap offsets as per-lane constants
  bf16x8 bw[2][TN];
  int koff[2][8];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    float wv[TN][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * ks + 8 * h + e;
      const bool kok = k < K;
      const int kk = kok ? k : 0;
      const int c = kk / 9, t = kk - 9 * c, ky = t / 3, kx = t - 3 * ky;
      koff[ks][e] = c * plane + ky * RS + kx;            // taps >= K read tap 0 (finite) against a zero weight
#pragma unroll
      for (int j = 0; j < TN; ++j) wv[j][e] = kok ? w[(int64_t)(32 * j + l31) * K + kk] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const uint4 pk = make_uint4(pack_bf16x2(wv[j][0], wv[j][1]), pack_bf16x2(wv[j][2], wv[j][3]),
                                  pack_bf16x2(wv[j][4], wv[j][5]), pack_bf16x2(wv[j][6], wv[j][7]));
      bw[ks][j] = __builtin_bit_cast(bf16x8, pk);
    }
  }
  float bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bv[j] = bias[32 * j + l31];
  __syncthreads();

  const int nwin = nwr * Wp, ntiles = (nwin + 7) / 8;
  const int64_t o0 = (int64_t)(b * Hp + py0) * Wp * Co;
  const int plane16 = Hp * Wp * 16;                     // elements of one 16-channel block of one image
  const int64_t o16 = ((int64_t)b * (Co / 16) * Hp + py0) * Wp * 16;
  const __amdgpu_buffer_rsrc_t rp = buf_rsrc(pooled16 + (C16 ? o16 : o0)), ra = buf_rsrc(amax + o0);
  for (int t = wave; t < ntiles; t += 4) {
    int wdx = 8 * t + (l31 >> 2);
    if (wdx >= nwin) wdx = 0;
    const int wr = (wdx >= Wp ? 1 : 0) + (wdx >= 2 * Wp ? 1 : 0) + (wdx >= 3 * Wp ? 1 : 0);
    const int px = wdx - __mul24(wr, Wp), j4 = l31 & 3;
    const float* ap = patch + __mul24(2 * wr + (j4 >> 1), RS) + 2 * px + (j4 & 1);
    f32x16 acc[TN];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ap[koff[ks][e]];
      const uint4 pk = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                  pack_bf16x2(v[6], v[7]));
      const bf16x8 a = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[ks][j], ks == 0 ? zero : acc[j], 0, 0, 0);
    }
    const bool inner = 8 * t + 8 <= nwin;
    if (C16) {
      // C16 + wide stores: the tile's 8 windows x Co channels go through a wave-private LDS scratch behind the patch --
      // pooled as [16-channel block][window][16] bf16, arg-max as [window][Co] bytes -- and leave as ONE 16-byte-per-lane
      // store each (the element-wise form issued 8 two-byte and 8 one-byte store instructions per tile and was bound by that)
      char* const scr = reinterpret_cast<char*>(patch) + CI * C0_PR * RS * 4 + wave * (8 * Co * 3);
      char* const sam = scr + 8 * Co * 2;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float best = acc[j][4 * g];
          int a = 0;
          if (acc[j][4 * g + 1] > best) { best = acc[j][4 * g + 1]; a = 1; }
          if (acc[j][4 * g + 2] > best) { best = acc[j][4 * g + 2]; a = 2; }
          if (acc[j][4 * g + 3] > best) { best = acc[j][4 * g + 3]; a = 3; }
          best += bv[j];
          const int win = 2 * g + h;
          *reinterpret_cast<uint16_t*>(scr + (((2 * j + (l31 >> 4)) * 8 + win) * 16 + (l31 & 15)) * 2) = bf16_bits(best > 0.f ? best : 0.f);
          *reinterpret_cast<uint8_t*>(sam + win * Co + 32 * j + l31) = best > 0.f ? (uint8_t)a : (uint8_t)4;
        }
      }
      asm volatile("" ::: "memory");        // the wave's LDS accesses execute in order; keep the compiler's order as well
      {
        const int byte = lane * 16;          // pooled: 2 TN blocks x 8 windows x 32 bytes = 8 * Co * 2 bytes
        if (byte < 8 * Co * 2) {
          const int blk = byte >> 8, win = (byte & 255) >> 5, inrun = byte & 31;
          const float4 v = *reinterpret_cast<const float4*>(scr + byte);
          const bool ok = inner || 8 * t + win < nwin;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rp,
                                                 ok ? (int)(blk * plane16 * 2 + (8 * t + win) * 32 + inrun) : (int)BUF_OOB, 0, 0);
        }
        if (byte < 8 * Co) {                 // arg-max: 8 windows x Co bytes, contiguous in NHWC
          const int win = byte / Co;
          const float4 v = *reinterpret_cast<const float4*>(sam + byte);
          const bool ok = inner || 8 * t + win < nwin;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ra, ok ? (int)(8 * t * Co + byte) : (int)BUF_OOB, 0, 0);
        }
      }
      asm volatile("" ::: "memory");
      continue;
    }
    const uint32_t vl = (uint32_t)__mul24(8 * t + h, Co) + (uint32_t)l31;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bool ok = inner || 8 * t + 2 * g + h < nwin;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float best = acc[j][4 * g];
        int a = 0;
        if (acc[j][4 * g + 1] > best) { best = acc[j][4 * g + 1]; a = 1; }
        if (acc[j][4 * g + 2] > best) { best = acc[j][4 * g + 2]; a = 2; }
        if (acc[j][4 * g + 3] > best) { best = acc[j][4 * g + 3]; a = 3; }
        best += bv[j];
        const uint32_t so = (uint32_t)(2 * g * Co + 32 * j);
        buf_store2(rp, bf16_bits(best > 0.f ? best : 0.f), ok ? 2u * vl : BUF_OOB, 2u * so);
        buf_store1(ra, best > 0.f ? (uint8_t)a : (uint8_t)4, ok ? vl : BUF_OOB, so);
      }
    }
  }
}

// ------------------------------------------------------------------ forward on bf16 MFMA, C16 output, persistent
// The kernel above is a one-shot workgroup: stage the patch (a chain of HBM loads, nothing else to do), barrier, compute --
// with the fp32 patch (56 KB at W = 448) only two workgroups fit a CU, and half of a workgroup's 40 us life was the staging
// (2.2 ms per launch at 512 x 448 x 448 against 1.1 ms of HBM time).  Here a PERSISTENT workgroup walks (image, block of
// C0_FR pooled rows) items; the next item's image rows are loaded into REGISTERS before the current item's tiles are
// computed and go to LDS after them, so HBM latency runs under the MFMA / epilogue work; the patch is held as bf16 (the
// rounding the MFMA operand gets anyway, applied once per pixel instead of once per tap): 28 KB; eight waves per workgroup
// (half the prefetch registers per thread), two workgroups per CU.
// XH: the image is __half (the dataset's features) / float.
// NWV waves per workgroup: 8 (two workgroups per CU) for __half images, 16 (one per CU) for float ones -- 16 waves per CU either
// way, and the prefetch stays at 16 VGPRs per thread (the budget at 4 waves per SIMD is 128).
template <int CI, int TN, bool XH>
__global__ __launch_bounds__(XH ? 512 : 1024, 4) void conv0_fwd_c16_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, uint16_t* pooled16, uint8_t* amax,
                                                            int B, int H, int W, int Hp, int Wp, int RS, int nblk) {
  extern __shared__ __attribute__((aligned(16))) char c0lds[];
  uint16_t* const patch = reinterpret_cast<uint16_t*>(c0lds);          // [CI][C0_PR][RS] bf16, RS % 64 == 16
  constexpr int K = 9 * CI, Co = 32 * TN;
  static_assert(K <= 32, "two 16-deep k-steps cover at most 32 taps");
  static_assert(CI * C0_PR <= 32, "the staged rows are dealt over 32 wave slots");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int plane = C0_PR * RS;
  const int W4 = W >> 2;
  constexpr int NWV = XH ? 8 : 16;
  constexpr int NQ = 2 * (32 / NWV);                     // pieces per thread: 32 / NWV rows x 2 column pieces (host check: W <= 512)
  typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
  h16x4 rh[XH ? NQ : 1];
  float4 rf[XH ? 1 : NQ];

  // weights as bf16 B fragments (element e of k-step ks: tap k = 16 ks + 8 h + e) parked in LDS, one 16-byte read per
  // fragment and tile (in registers they were 8 TN VGPRs the prefetch needs); tap offsets as per-lane constants, two 16-bit
  // offsets per register
  char* const wl = c0lds + CI * C0_PR * RS * 2 + NWV * (8 * Co * 3);     // [2 ks][TN][64 lanes][16 bytes]
  uint32_t koff2[2][4];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    float wv[TN][8];
    int ko[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * ks + 8 * h + e;
      const bool kok = k < K;
      const int kk = kok ? k : 0;
      const int c = kk / 9, t = kk - 9 * c, ky = t / 3, kx = t - 3 * ky;
      ko[e] = c * plane + ky * RS + kx;                  // taps >= K read tap 0 (finite) against a zero weight
#pragma unroll
      for (int j = 0; j < TN; ++j) wv[j][e] = kok ? w[(int64_t)(32 * j + l31) * K + kk] : 0.f;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) koff2[ks][m] = (uint32_t)ko[2 * m] | ((uint32_t)ko[2 * m + 1] << 16);
    if (wave == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
        *reinterpret_cast<uint4*>(wl + ((ks * TN + j) * 64 + lane) * 16) =
            make_uint4(pack_bf16x2(wv[j][0], wv[j][1]), pack_bf16x2(wv[j][2], wv[j][3]), pack_bf16x2(wv[j][4], wv[j][5]),
                       pack_bf16x2(wv[j][6], wv[j][7]));
    }
  }
  float bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bv[j] = bias[32 * j + l31];

  // wave wv stages rows wv, wv + NWV, .. of the CI * C0_PR patch rows; a lane takes the 4-pixel pieces lane and lane + 64 of a row
  // (W <= 512): row bases are wave-uniform, the prefetch registers hold data only
  auto load_item = [&](int item) {
    const int b = item / nblk, py0 = C0_FR * (item - b * nblk);
    const int nrows = 2 * min(C0_FR, Hp - py0) + 2;
    const int64_t img = ((int64_t)b * CI * H + 2 * py0) * W;
#pragma unroll
    for (int i = 0; i < NQ / 2; ++i) {
      const int r = wave + NWV * i, c = r / C0_PR, rr = r - c * C0_PR;
      const int64_t e = img + ((int64_t)c * H + rr) * W;
      const bool rok = r < CI * C0_PR && rr < nrows;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c4 = lane + 64 * u;
        const bool ok = rok && c4 < W4;
        if (XH) rh[2 * i + u] = ok ? *reinterpret_cast<const h16x4*>(static_cast<const uint16_t*>(x) + e + 4 * c4) : h16x4{0, 0, 0, 0};
        else rf[2 * i + u] = ok ? *reinterpret_cast<const float4*>(static_cast<const float*>(x) + e + 4 * c4) : f4zero();
      }
    }
  };
  auto store_item = [&]() {
#pragma unroll
    for (int i = 0; i < NQ / 2; ++i) {
      const int r = wave + NWV * i;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c4 = lane + 64 * u;
        if (r < CI * C0_PR && c4 < W4) {
          float4 v;
          if (XH) v = make_float4((float)rh[2 * i + u][0], (float)rh[2 * i + u][1], (float)rh[2 * i + u][2], (float)rh[2 * i + u][3]);
          else v = rf[2 * i + u];
          *reinterpret_cast<uint2*>(patch + r * RS + 4 * c4) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
        }
      }
    }
  };

  const int nitems = B * nblk;
  int item = blockIdx.x;
  if (item >= nitems) return;
  load_item(item);
  char* const scr = c0lds + CI * C0_PR * RS * 2 + wave * (8 * Co * 3);
  char* const sam = scr + 8 * Co * 2;
  const int plane16 = Hp * Wp * 16;                     // elements of one 16-channel block of one image
  for (; item < nitems; item += gridDim.x) {
    __syncthreads();                                    // the previous item's tiles are done with the patch
    store_item();
    __syncthreads();
    const int nxt = item + gridDim.x;
    if (nxt < nitems) load_item(nxt);                   // in flight under this item's tiles
    const int b = item / nblk, py0 = C0_FR * (item - b * nblk);
    const int nwr = min(C0_FR, Hp - py0);
    const int nwin = nwr * Wp, ntiles = (nwin + 7) / 8;
    const int64_t o0 = (int64_t)(b * Hp + py0) * Wp * Co;
    const int64_t o16 = ((int64_t)b * (Co / 16) * Hp + py0) * Wp * 16;
    const __amdgpu_buffer_rsrc_t rp = buf_rsrc(pooled16 + o16), ra = buf_rsrc(amax + o0);
    for (int t = wave; t < ntiles; t += NWV) {
      int wdx = 8 * t + (l31 >> 2);
      if (wdx >= nwin) wdx = 0;
      const int wr = (wdx >= Wp ? 1 : 0) + (wdx >= 2 * Wp ? 1 : 0) + (wdx >= 3 * Wp ? 1 : 0);
      const int px = wdx - __mul24(wr, Wp), j4 = l31 & 3;
      const uint16_t* ap = patch + __mul24(2 * wr + (j4 >> 1), RS) + 2 * px + (j4 & 1);
      f32x16 acc[TN];
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint32_t pkv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
          pkv[m] = (uint32_t)ap[koff2[ks][m] & 0xffffu] | ((uint32_t)ap[koff2[ks][m] >> 16] << 16);
        const bf16x8 a = __builtin_bit_cast(bf16x8, make_uint4(pkv[0], pkv[1], pkv[2], pkv[3]));
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const bf16x8 bwf = *reinterpret_cast<const bf16x8*>(wl + ((ks * TN + j) * 64 + lane) * 16);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bwf, ks == 0 ? zero : acc[j], 0, 0, 0);
        }
      }
      const bool inner = 8 * t + 8 <= nwin;
      // the tile's 8 windows x Co channels through the wave's LDS scratch -- pooled as [16-channel block][window][16] bf16,
      // arg-max as [window][Co] bytes -- and out as ONE 16-byte-per-lane store each
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float best = acc[j][4 * g];
          int a = 0;
          if (acc[j][4 * g + 1] > best) { best = acc[j][4 * g + 1]; a = 1; }
          if (acc[j][4 * g + 2] > best) { best = acc[j][4 * g + 2]; a = 2; }
          if (acc[j][4 * g + 3] > best) { best = acc[j][4 * g + 3]; a = 3; }
          best += bv[j];
          const int win = 2 * g + h;
          *reinterpret_cast<uint16_t*>(scr + (((2 * j + (l31 >> 4)) * 8 + win) * 16 + (l31 & 15)) * 2) = bf16_bits(best > 0.f ? best : 0.f);
          *reinterpret_cast<uint8_t*>(sam + win * Co + 32 * j + l31) = best > 0.f ? (uint8_t)a : (uint8_t)4;
        }
      }
      asm volatile("" ::: "memory");        // the wave's LDS accesses execute in order; keep the compiler's order as well
      {
        const int byte = lane * 16;          // pooled: 2 TN blocks x 8 windows x 32 bytes = 8 * Co * 2 bytes
        if (byte < 8 * Co * 2) {
          const int blk = byte >> 8, win = (byte & 255) >> 5, inrun = byte & 31;
          const float4 v = *reinterpret_cast<const float4*>(scr + byte);
          const bool ok = inner || 8 * t + win < nwin;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rp,
                                                 ok ? (int)(blk * plane16 * 2 + (8 * t + win) * 32 + inrun) : (int)BUF_OOB, 0, 0);
        }
        if (byte < 8 * Co) {                 // arg-max: 8 windows x Co bytes, contiguous in NHWC
          const int win = byte / Co;
          const float4 v = *reinterpret_cast<const float4*>(sam + byte);
          const bool ok = inner || 8 * t + win < nwin;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ra, ok ? (int)(8 * t * Co + byte) : (int)BUF_OOB, 0, 0);
        }
      }
      asm volatile("" ::: "memory");
    }
  }
}

// ------------------------------------------------------------------ wgrad
// persistent grid; 256 threads; LDS = CI*PLANE (x patch, 4 rows) + Wp*Co (dP row) floats + Wp*Co bytes (arg-max row)
template <int CI, int TN>
__global__ __launch_bounds__(256) void conv0_wgrad_kernel(const void* __restrict__ x, int xh, const float* __restrict__ dp,
                                                          const uint8_t* __restrict__ am, float* slab, float* bias_slab,
                                                          int B, int H, int W, int Hp, int Wp, int RS, int PLANE) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int K = 9 * CI, Co = 32 * TN;
  float* patch = lds;
  float* dps = lds + ((CI * PLANE + 3) & ~3);
  uint8_t* ams = reinterpret_cast<uint8_t*>(dps + Wp * Co);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  // lane l31 owns tap i = l31 of the 32-row A operand (rows >= K are don't-care: never written out)
  const int i = l31 < K ? l31 : 0;
  const int c = i / 9, t9 = i - 9 * c, ky = t9 / 3, kx = t9 - 3 * ky;
  const float* ap = patch + c * PLANE + ky * RS + kx + h;   // + h: the pixel pair (dx = h) of one MFMA
  f32x16 acc[TN];
  float bsum[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    bsum[j] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  }
  const int rows_total = B * Hp;
  const int rowv = Wp * Co;   // floats in one pooled-gradient row
  for (int row = blockIdx.x; row < rows_total; row += gridDim.x) {
    const int b = row / Hp, py = row - b * Hp;
    __syncthreads();   // previous row fully consumed
    for (int e = tid; e < CI * 4 * (W / 4); e += 256) {
      const int c4 = e % (W / 4);
      const int r = (e / (W / 4)) & 3;
      const int cc = e / (W / 4) / 4;
      const float4 v = c0_load4(x, ((int64_t)(b * CI + cc) * H + 2 * py + r) * W + 4 * c4, xh);
      float* d = patch + cc * PLANE + r * RS + 4 * c4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    const float* dprow = dp + (int64_t)row * rowv;
    const uint8_t* amrow = am + (int64_t)row * rowv;
    for (int e = tid; e < rowv / 4; e += 256) {
      reinterpret_cast<float4*>(dps)[e] = reinterpret_cast<const float4*>(dprow)[e];
      reinterpret_cast<uint32_t*>(ams)[e] = reinterpret_cast<const uint32_t*>(amrow)[e];
    }
    __syncthreads();
    for (int px = wave; px < Wp; px += 4) {
      float d[TN];
      int id[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) { d[j] = dps[px * Co + 32 * j + l31]; id[j] = ams[px * Co + 32 * j + l31]; }
#pragma unroll
      for (int hs = 0; hs < 2; ++hs) {          // pixel pair (dy = hs, dx = h)
        const float a = ap[hs * RS + 2 * px];
        const int jj = 2 * hs + h;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float bvv = id[j] == jj ? d[j] : 0.f;
          bsum[j] += bvv;
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bvv, acc[j], 0, 0, 0);
        }
      }
    }
  }
  // combine the 4 waves of the workgroup through LDS, then ONE partial per workgroup
  __syncthreads();
  float* comb = lds;                               // [4][32][Co] floats, fits the staging area
  float* cb = lds + 4 * 32 * Co;                   // [4][Co]
#pragma unroll
  for (int j = 0; j < TN; ++j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
      comb[(wave * 32 + k) * Co + 32 * j + l31] = acc[j][r];
    }
    const float s = bsum[j] + __shfl_xor(bsum[j], 32, 64);
    if (h == 0) cb[wave * Co + 32 * j + l31] = s;
  }
  __syncthreads();
  float* out = slab + (int64_t)blockIdx.x * 32 * Co;
  for (int e = tid; e < 32 * Co; e += 256)
    out[e] = comb[e] + comb[32 * Co + e] + comb[2 * 32 * Co + e] + comb[3 * 32 * Co + e];
  for (int e = tid; e < Co; e += 256)
    bias_slab[(int64_t)blockIdx.x * Co + e] = cb[e] + cb[Co + e] + cb[2 * Co + e] + cb[3 * Co + e];
}

// ------------------------------------------------------------------ wgrad, next row prefetched into registers (round 3)
// The kernel above is load -> barrier -> compute -> barrier per pooled row; with three workgroups per CU the HBM round trip of a
// row's operands still stood in front of its MFMAs.  Here the NEXT row's image rows, pooled gradient and arg-max bytes are loaded
// into registers before the current row is computed and go to LDS after it (W <= 256, Wp * Co <= 8192; wider images take the
// kernel above).  XH: the image is __half / float.
template <int CI, int TN, bool XH>
__global__ __launch_bounds__(256) void conv0_wgrad_pf_kernel(const void* __restrict__ x, const float* __restrict__ dp,
                                                          const uint8_t* __restrict__ am, float* slab, float* bias_slab,
                                                          int B, int H, int W, int Hp, int Wp, int RS, int PLANE) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int K = 9 * CI, Co = 32 * TN;
  float* patch = lds;
  float* dps = lds + ((CI * PLANE + 3) & ~3);
  uint8_t* ams = reinterpret_cast<uint8_t*>(dps + Wp * Co);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  // lane l31 owns tap i = l31 of the 32-row A operand (rows >= K are don't-care: never written out)
  const int i = l31 < K ? l31 : 0;
  const int c = i / 9, t9 = i - 9 * c, ky = t9 / 3, kx = t9 - 3 * ky;
  const float* ap = patch + c * PLANE + ky * RS + kx + h;   // + h: the pixel pair (dx = h) of one MFMA
  f32x16 acc[TN];
  float bsum[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    bsum[j] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  }
  const int rows_total = B * Hp;
  const int rowv = Wp * Co;   // floats in one pooled-gradient row
  // prefetch registers: image = (channel: round, image row: wave, 4-pixel piece: lane), pooled gradient / arg-max = 16-byte /
  // 4-byte pieces tid + 256 i
  constexpr int DR = 8;
  typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
  h16x4 xh_[XH ? CI : 1];
  float4 xf_[XH ? 1 : CI];
  float4 dr_[DR];
  uint32_t ar_[DR];
  const int W4 = W >> 2, npieces = rowv >> 2;
  auto load_row = [&](int row) {
    const int b = row / Hp, py = row - b * Hp;
#pragma unroll
    for (int cc = 0; cc < CI; ++cc) {
      const int64_t e = ((int64_t)(b * CI + cc) * H + 2 * py + wave) * W + 4 * lane;
      const bool ok = lane < W4;
      if (XH) xh_[cc] = ok ? *reinterpret_cast<const h16x4*>(static_cast<const uint16_t*>(x) + e) : h16x4{0, 0, 0, 0};
      else xf_[cc] = ok ? *reinterpret_cast<const float4*>(static_cast<const float*>(x) + e) : f4zero();
    }
    const float* dprow = dp + (int64_t)row * rowv;
    const uint8_t* amrow = am + (int64_t)row * rowv;
#pragma unroll
    for (int i = 0; i < DR; ++i) {
      const int e = tid + 256 * i;
      const bool in = e < npieces;
      dr_[i] = in ? reinterpret_cast<const float4*>(dprow)[e] : f4zero();
      ar_[i] = in ? reinterpret_cast<const uint32_t*>(amrow)[e] : 0u;
    }
  };
  auto store_row = [&]() {
    if (lane < W4) {
#pragma unroll
      for (int cc = 0; cc < CI; ++cc) {
        float4 v;
        if (XH) v = make_float4((float)xh_[cc][0], (float)xh_[cc][1], (float)xh_[cc][2], (float)xh_[cc][3]);
        else v = xf_[cc];
        float* d = patch + cc * PLANE + wave * RS + 4 * lane;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    }
#pragma unroll
    for (int i = 0; i < DR; ++i) {
      const int e = tid + 256 * i;
      if (e < npieces) {
        reinterpret_cast<float4*>(dps)[e] = dr_[i];
        reinterpret_cast<uint32_t*>(ams)[e] = ar_[i];
      }
    }
  };
  int row = blockIdx.x;
  if (row < rows_total) load_row(row);
  for (; row < rows_total; row += gridDim.x) {
    __syncthreads();   // previous row fully consumed
    store_row();
    __syncthreads();
    if (row + (int)gridDim.x < rows_total) load_row(row + gridDim.x);     // in flight under this row's MFMAs
    for (int px = wave; px < Wp; px += 4) {
      float d[TN];
      int id[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) { d[j] = dps[px * Co + 32 * j + l31]; id[j] = ams[px * Co + 32 * j + l31]; }
#pragma unroll
      for (int hs = 0; hs < 2; ++hs) {          // pixel pair (dy = hs, dx = h)
        const float a = ap[hs * RS + 2 * px];
        const int jj = 2 * hs + h;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float bvv = id[j] == jj ? d[j] : 0.f;
          bsum[j] += bvv;
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bvv, acc[j], 0, 0, 0);
        }
      }
    }
  }
  // combine the 4 waves of the workgroup through LDS, then ONE partial per workgroup
  __syncthreads();
  float* comb = lds;                               // [4][32][Co] floats, fits the staging area
  float* cb = lds + 4 * 32 * Co;                   // [4][Co]
#pragma unroll
  for (int j = 0; j < TN; ++j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
      comb[(wave * 32 + k) * Co + 32 * j + l31] = acc[j][r];
    }
    const float s = bsum[j] + __shfl_xor(bsum[j], 32, 64);
    if (h == 0) cb[wave * Co + 32 * j + l31] = s;
  }
  __syncthreads();
  float* out = slab + (int64_t)blockIdx.x * 32 * Co;
  for (int e = tid; e < 32 * Co; e += 256)
    out[e] = comb[e] + comb[32 * Co + e] + comb[2 * 32 * Co + e] + comb[3 * 32 * Co + e];
  for (int e = tid; e < Co; e += 256)
    bias_slab[(int64_t)blockIdx.x * Co + e] = cb[e] + cb[Co + e] + cb[2 * Co + e] + cb[3 * Co + e];
}

// ------------------------------------------------------------------ wgrad on bf16 MFMA (bf16 path, configs[3])
// dW[tap][co] = sum over conv-output pixels of x(pixel, tap) * dY(pixel, co): the reduction index (pixels) is the MFMA k.
//   A (rows = the 27 taps, k = 8 consecutive pixels of a conv row): the image rows are staged as bf16 in THREE copies
//     shifted by kx = 0, 1, 2, so that a lane's 8 pixels of tap (c, ky, kx) are one aligned ds_read_b128;
//   B (k = the same 8 pixels, cols = co): rebuilt per lane from the 4 pool windows the pixels fall into (bf16 pooled
//     gradient + arg-max byte, staged per pooled row): pixel (hs, 2m + parity) of window m keeps the gradient iff
//     arg-max == 2 hs + parity; the 4 windows serve both conv rows hs = 0, 1 of the pooled row.
// Persistent workgroups walk pooled rows; each wave keeps 32 x Co partial sums in accumulators for its whole life and
// the four waves combine through LDS at the end: ONE slab per workgroup (deterministic, no atomics).
// The NEXT row's operands (image rows, pooled gradient, arg-max bytes) are loaded into registers before the current row is
// computed and go to LDS after it: HBM latency runs under the MFMA / routing work instead of in front of it (the one-shot
// form -- load, barrier, compute, barrier -- spent two thirds of a row's 21 us waiting at 2 workgroups per CU).
// XH: the image is __half / float.  W <= 512 (one 8-pixel chunk per lane and image row), Wp * Co <= 16384.
template <int CI, int TN, bool XH>
__global__ __launch_bounds__(256, 2) void conv0_wgrad_bf16_kernel(const void* __restrict__ x, const uint16_t* __restrict__ dp,
                                                                  const uint8_t* __restrict__ am, float* slab,
                                                                  float* bias_slab, int B, int H, int W, int Hp, int Wp,
                                                                  int RSTR, int NG) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int K = 9 * CI, Co = 32 * TN;
  static_assert(K <= 32, "the taps are the 32 rows of one MFMA A operand");
  const int WpP = 8 * NG;                                   // windows incl. padding (16 pixels = 8 windows per group)
  char* const P16 = reinterpret_cast<char*>(lds);           // [3 kx][CI][4 rows][RSTR bytes]
  uint16_t* const dps = reinterpret_cast<uint16_t*>(P16 + 3 * CI * 4 * RSTR);      // [WpP][Co] bf16
  uint8_t* const ams = reinterpret_cast<uint8_t*>(dps + WpP * Co);                  // [WpP][Co]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int tap = l31 < K ? l31 : 0;                        // rows >= K: any finite data, never written out
  const int c = tap / 9, t9 = tap - 9 * c, ky = t9 / 3, kx = t9 - 3 * ky;
  const char* const arow = P16 + ((kx * CI + c) * 4 + ky) * RSTR + 16 * h;   // + hs * RSTR + 32 * g
  f32x16 acc[TN];
  float bsum[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    bsum[j] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  }
  const int rows_total = B * Hp;
  const int rowv = Wp * Co;
  // prefetch registers: image = (channel cc: round, image row: wave, 8-pixel chunk: lane), 12 pixels each (8 + the 4 the
  // shifted copies need); pooled gradient / arg-max = 16-byte / 8-byte pieces tid + 256 i
  constexpr int DR = 8;
  typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
  h16x4 xh_[XH ? CI : 1][3];
  float4 xf_[XH ? 1 : CI][3];
  uint4 dr_[DR];
  uint2 ar_[DR];
  const int npieces = WpP * Co / 8;
  auto load_row = [&](int row) {
    const int b = row / Hp, py = row - b * Hp;
#pragma unroll
    for (int cc = 0; cc < CI; ++cc) {
      const int64_t src = ((int64_t)(b * CI + cc) * H + 2 * py + wave) * W + 8 * lane;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const bool ok = 8 * lane + 4 * q < W;                       // W % 4 == 0
        if (XH) xh_[cc][q] = ok ? *reinterpret_cast<const h16x4*>(static_cast<const uint16_t*>(x) + src + 4 * q) : h16x4{0, 0, 0, 0};
        else xf_[cc][q] = ok ? *reinterpret_cast<const float4*>(static_cast<const float*>(x) + src + 4 * q) : f4zero();
      }
    }
    const uint16_t* dprow = dp + (int64_t)row * rowv;
    const uint8_t* amrow = am + (int64_t)row * rowv;
#pragma unroll
    for (int i = 0; i < DR; ++i) {
      const int e = tid + 256 * i;
      const bool in = 8 * e < rowv;                      // Co % 8 == 0: a piece never straddles the row end
      dr_[i] = in ? reinterpret_cast<const uint4*>(dprow)[e] : make_uint4(0u, 0u, 0u, 0u);
      ar_[i] = in ? reinterpret_cast<const uint2*>(amrow)[e] : make_uint2(0x04040404u, 0x04040404u);   // padding windows: dead
    }
  };
  auto store_row = [&]() {
    if (lane < 2 * NG) {
#pragma unroll
      for (int cc = 0; cc < CI; ++cc) {
        float v[12];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          if (XH) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[4 * q + u] = (float)xh_[cc][q][u];
          } else {
            v[4 * q] = xf_[cc][q].x; v[4 * q + 1] = xf_[cc][q].y; v[4 * q + 2] = xf_[cc][q].z; v[4 * q + 3] = xf_[cc][q].w;
          }
        }
#pragma unroll
        for (int sft = 0; sft < 3; ++sft) {
          const uint4 pk = make_uint4(pack_bf16x2(v[sft], v[sft + 1]), pack_bf16x2(v[sft + 2], v[sft + 3]),
                                      pack_bf16x2(v[sft + 4], v[sft + 5]), pack_bf16x2(v[sft + 6], v[sft + 7]));
          *reinterpret_cast<uint4*>(P16 + ((sft * CI + cc) * 4 + wave) * RSTR + 16 * lane) = pk;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < DR; ++i) {
      const int e = tid + 256 * i;
      if (e < npieces) {
        reinterpret_cast<uint4*>(dps)[e] = dr_[i];
        reinterpret_cast<uint2*>(ams)[e] = ar_[i];
      }
    }
  };
  int row = blockIdx.x;
  if (row < rows_total) load_row(row);
  for (; row < rows_total; row += gridDim.x) {
    __syncthreads();   // previous row fully consumed
    store_row();
    __syncthreads();
    if (row + (int)gridDim.x < rows_total) load_row(row + gridDim.x);     // in flight under this row's MFMAs
    for (int g = wave; g < NG; g += 4) {
      // routing without branches or selects: a window's gradient, zeroed when the window is dead (code 4), shifted to the
      // 16-bit field of its arg-max pixel in a 64-bit word [pixel 0 | 1 | 2 | 3] -- the low half is the B-operand pair of
      // conv row hs = 0, the high half that of hs = 1 (the compare-and-select form compiled to ~60 skip branches per group)
      uint32_t wlo[TN][4], whi[TN][4];
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int o = (8 * g + 4 * h + m) * Co + 32 * j + l31;
          const uint32_t code = ams[o];
          const uint32_t dz = (uint32_t)dps[o] & ((code >> 2) - 1u);          // codes 0..3 keep, 4 clears
          bsum[j] += __uint_as_float(dz << 16);
          const uint64_t f = (uint64_t)dz << ((code & 3u) * 16u);
          wlo[j][m] = (uint32_t)f;
          whi[j][m] = (uint32_t)(f >> 32);
        }
#pragma unroll
      for (int hs = 0; hs < 2; ++hs) {
        const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(arow + hs * RSTR + 32 * g));
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const bf16x8 bf = __builtin_bit_cast(bf16x8, hs ? make_uint4(whi[j][0], whi[j][1], whi[j][2], whi[j][3])
                                                          : make_uint4(wlo[j][0], wlo[j][1], wlo[j][2], wlo[j][3]));
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[j], 0, 0, 0);
        }
      }
    }
  }
  // combine the 4 waves of the workgroup through LDS, then ONE partial per workgroup
  __syncthreads();
  float* comb = lds;                               // [4][32][Co] floats
  float* cb = lds + 4 * 32 * Co;                   // [4][Co]
#pragma unroll
  for (int j = 0; j < TN; ++j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
      comb[(wave * 32 + k) * Co + 32 * j + l31] = acc[j][r];
    }
    const float sv = bsum[j] + __shfl_xor(bsum[j], 32, 64);
    if (h == 0) cb[wave * Co + 32 * j + l31] = sv;
  }
  __syncthreads();
  float* out = slab + (int64_t)blockIdx.x * 32 * Co;
  for (int e = tid; e < 32 * Co; e += 256)
    out[e] = comb[e] + comb[32 * Co + e] + comb[2 * 32 * Co + e] + comb[3 * 32 * Co + e];
  for (int e = tid; e < Co; e += 256)
    bias_slab[(int64_t)blockIdx.x * Co + e] = cb[e] + cb[Co + e] + cb[2 * Co + e] + cb[3 * Co + e];
}

// slab[parts][32][Co] -> dw[co][k]; bias_slab[parts][Co] -> dbias[co].  grid = K + 1 blocks of 256 threads.
__global__ void conv0_wgrad_reduce_kernel(const float* slab, const float* bias_slab, float* dw, float* dbias,
                                          int parts, int K, int Co) {
  __shared__ float red[256];
  const int k = blockIdx.x;          // k == K: the bias row
  const int co = threadIdx.x % Co, sl = threadIdx.x / Co, nsl = 256 / Co;
  float v = 0.f;
  if (sl < nsl)
    for (int p = sl; p < parts; p += nsl)
      v += (k < K) ? slab[((int64_t)p * 32 + k) * Co + co] : bias_slab[(int64_t)p * Co + co];
  red[threadIdx.x] = v;
  __syncthreads();
  if (sl == 0) {
    for (int s2 = 1; s2 < nsl; ++s2) v += red[s2 * Co + co];
    if (k < K) dw[(int64_t)co * K + k] = v; else dbias[co] = v;
  }
}

static bool c0_supported(int Ci, int H, int W, int Co, int stride) {
  const int Hp = (H - 2) / 2, Wp = (W - 2) / 2;
  // wgrad maps the 9*Ci taps onto the 32 rows of one MFMA A operand: Ci <= 3
  if (!(Ci >= 1 && Ci <= 3 && stride == 1 && (Co == 32 || Co == 64) && W % 4 == 0 && H >= 6 && Hp > 0 && Wp > 0))
    return false;
  const size_t fwd = (size_t)Ci * C0_PR * c0_round_stride(W, 16) * 4 + (size_t)4 * 8 * Co * 5;
  const int rs = c0_round_stride(W, 11);
  int plane = 4 * rs;
  while (plane % 32 != 3) ++plane;
  const size_t wg = ((size_t)Ci * plane + 4) * 4 + (size_t)Wp * Co * 5;
  return fwd <= 160 * 1024 && wg <= 160 * 1024;      // one workgroup's LDS (the launchers raise the 64 KB default)
}

constexpr int kC0Blocks = 768;   // persistent wgrad grid: 3 workgroups per CU

}  // namespace vqa

using namespace vqa;

#define C0_DISPATCH(CI, TN, ...)                                                          \
  switch ((CI) * 10 + (TN)) {                                                             \
    case 11: { constexpr int kCI = 1, kTN = 1; __VA_ARGS__; } break;                      \
    case 12: { constexpr int kCI = 1, kTN = 2; __VA_ARGS__; } break;                      \
    case 21: { constexpr int kCI = 2, kTN = 1; __VA_ARGS__; } break;                      \
    case 22: { constexpr int kCI = 2, kTN = 2; __VA_ARGS__; } break;                      \
    case 31: { constexpr int kCI = 3, kTN = 1; __VA_ARGS__; } break;                      \
    case 32: { constexpr int kCI = 3, kTN = 2; __VA_ARGS__; } break;                      \
    case 41: { constexpr int kCI = 4, kTN = 1; __VA_ARGS__; } break;                      \
    case 42: { constexpr int kCI = 4, kTN = 2; __VA_ARGS__; } break;                      \
    default: set_error("conv0: unsupported Ci=%d Co=%d", CI, 32 * (TN)); return VQA_ERR_INVALID; \
  }

extern "C" {

int vqa_conv0_supported(int Ci, int H, int W, int Co, int stride) { return c0_supported(Ci, H, W, Co, stride) ? 1 : 0; }

int vqa_conv0_relu_pool_fwd(const void* x_nchw, int x_is_fp16, const float* w, const float* bias, void* pooled, int pooled_is_bf16,
                            uint8_t* argmax, int B, int Ci, int H, int W, int Co, vqa_stream_t stream) {
  VQA_REQUIRE(x_nchw && w && bias && pooled && argmax && B > 0, "vqa_conv0_relu_pool_fwd: bad args");
  VQA_REQUIRE(c0_supported(Ci, H, W, Co, 1), "vqa_conv0_relu_pool_fwd: unsupported shape Ci=%d H=%d W=%d Co=%d", Ci, H, W, Co);
  VQA_REQUIRE(((uintptr_t)x_nchw % 16) == 0, "vqa_conv0_relu_pool_fwd: input must be 16-byte aligned");
  const int xh = x_is_fp16 ? 1 : 0;
  const int Hp = (H - 2) / 2, Wp = (W - 2) / 2, RS = c0_round_stride(W, 16);
  const size_t lds = (size_t)Ci * C0_PR * RS * 4;
  const dim3 grid((Hp + C0_FR - 1) / C0_FR, B);
#define C0_FWD_LAUNCH(OB)                                                                                              \
  C0_DISPATCH(Ci, Co / 32, {                                                                                           \
    auto kern = conv0_fwd_kernel<kCI, kTN, OB>;                                                                        \
    const size_t ldsk = lds + ((OB) == 0 ? (size_t)4 * 8 * Co * 5 : 0);   /* fp32 output: + the waves' store scratch */    \
    int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)ldsk, "attr(conv0_fwd)");                      \
    if (rc0) return rc0;                                                                                               \
    hipLaunchKernelGGL(kern, grid, dim3(256), ldsk, (hipStream_t)stream, x_nchw, xh, w, bias, pooled, argmax, H, W, Hp, Wp, RS); \
  })
  if (pooled_is_bf16 == 2 || pooled_is_bf16 == 4) {      // bf16 MFMA (image and weights rounded to bf16), bf16 output (4: C16)
    VQA_REQUIRE(Ci <= 3, "vqa_conv0_relu_pool_fwd: the bf16-MFMA first block needs Ci <= 3");
    VQA_REQUIRE((int64_t)Hp * Wp * Co * 2 < (1LL << 31), "vqa_conv0_relu_pool_fwd: one pooled image reaches 2 GiB");
    C0_DISPATCH(Ci, Co / 32, {
      if constexpr (kCI <= 3) {
        if (pooled_is_bf16 == 4) {
          // persistent, register-prefetched, bf16 patch (row stride = 16 mod 64 elements: the two image rows of a gather sit
          // 8 banks apart)
          int RS16 = W + 2;
          while (RS16 % 64 != 16) ++RS16;
          const int nwv = xh ? 8 : 16;
          const size_t lds16 = (size_t)kCI * C0_PR * RS16 * 2 + (size_t)nwv * 8 * Co * 3 + (size_t)2 * kTN * 1024;
          VQA_REQUIRE(W <= 512, "vqa_conv0_relu_pool_fwd: image too wide for the C16 kernel (W=%d)", W);
          const int nblk = (Hp + C0_FR - 1) / C0_FR;
          int blocks = 256 * (xh ? 2 : 1);
          if (blocks > B * nblk) blocks = B * nblk;
          if (xh) {
            auto kern = conv0_fwd_c16_kernel<kCI, kTN, true>;
            int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)lds16, "attr(conv0_fwd_c16)");
            if (rc0) return rc0;
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds16, (hipStream_t)stream, x_nchw, w, bias, static_cast<uint16_t*>(pooled),
                               argmax, B, H, W, Hp, Wp, RS16, nblk);
          } else {
            auto kern = conv0_fwd_c16_kernel<kCI, kTN, false>;
            int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)lds16, "attr(conv0_fwd_c16)");
            if (rc0) return rc0;
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds16, (hipStream_t)stream, x_nchw, w, bias, static_cast<uint16_t*>(pooled),
                               argmax, B, H, W, Hp, Wp, RS16, nblk);
          }
        } else {
          auto kern = conv0_fwd_bf16_kernel<kCI, kTN, false>;
          int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)lds, "attr(conv0_fwd_bf16)");
          if (rc0) return rc0;
          hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, x_nchw, xh, w, bias, static_cast<uint16_t*>(pooled),
                             argmax, H, W, Hp, Wp, RS);
        }
      }
    });
  } else if (pooled_is_bf16 == 3) { C0_FWD_LAUNCH(2); } else if (pooled_is_bf16) { C0_FWD_LAUNCH(1); } else { C0_FWD_LAUNCH(0); }
#undef C0_FWD_LAUNCH
  return check_hip(hipGetLastError(), "conv0_fwd launch");
}

int64_t vqa_conv0_wgrad_workspace_bytes(int Co) { return (int64_t)kC0Blocks * (32 + 1) * Co * 4; }

int vqa_conv0_wgrad(const void* x_nchw, int x_is_fp16, const float* dpooled, const uint8_t* argmax, float* dw, float* dbias, int B,
                    int Ci, int H, int W, int Co, float* workspace, int64_t workspace_bytes, vqa_stream_t stream) {
  VQA_REQUIRE(x_nchw && dpooled && argmax && dw && dbias && workspace, "vqa_conv0_wgrad: null pointer");
  VQA_REQUIRE(c0_supported(Ci, H, W, Co, 1), "vqa_conv0_wgrad: unsupported shape Ci=%d H=%d W=%d Co=%d", Ci, H, W, Co);
  if (workspace_bytes < vqa_conv0_wgrad_workspace_bytes(Co)) {
    set_error("vqa_conv0_wgrad: workspace too small");
    return VQA_ERR_WORKSPACE;
  }
  const int Hp = (H - 2) / 2, Wp = (W - 2) / 2, RS = c0_round_stride(W, 11);
  int PLANE = 4 * RS;
  while (PLANE % 32 != 3) ++PLANE;
  size_t lds = (((size_t)Ci * PLANE + 3) & ~(size_t)3) * 4 + (size_t)Wp * Co * 5;
  if (lds < (size_t)(4 * 32 + 4) * Co * 4) lds = (size_t)(4 * 32 + 4) * Co * 4;   // the end-of-kernel combine area
  int per_cu = (int)((160 * 1024) / lds);            // resident workgroups per CU by LDS (wide images need > 53 KB)
  if (per_cu > 3) per_cu = 3;
  if (per_cu < 1) per_cu = 1;
  int blocks = 256 * per_cu;
  if (blocks > B * Hp) blocks = B * Hp;
  float* slab = workspace;
  float* bias_slab = workspace + (int64_t)kC0Blocks * 32 * Co;
  hipStream_t s = (hipStream_t)stream;
  const bool prefetch = W <= 256 && Wp * Co <= 8192;     // the register-prefetching form (conv0_wgrad_pf_kernel)
  C0_DISPATCH(Ci, Co / 32, {
    if (prefetch && x_is_fp16) {
      auto kern = conv0_wgrad_pf_kernel<kCI, kTN, true>;
      int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)lds, "attr(conv0_wgrad)");
      if (rc0) return rc0;
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, x_nchw, dpooled, argmax, slab, bias_slab, B, H, W, Hp, Wp, RS, PLANE);
    } else if (prefetch) {
      auto kern = conv0_wgrad_pf_kernel<kCI, kTN, false>;
      int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)lds, "attr(conv0_wgrad)");
      if (rc0) return rc0;
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, x_nchw, dpooled, argmax, slab, bias_slab, B, H, W, Hp, Wp, RS, PLANE);
    } else {
      auto kern = conv0_wgrad_kernel<kCI, kTN>;
      int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)lds, "attr(conv0_wgrad)");
      if (rc0) return rc0;
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, x_nchw, x_is_fp16 ? 1 : 0, dpooled, argmax, slab, bias_slab, B, H, W,
                         Hp, Wp, RS, PLANE);
    }
  });
  int rc = check_hip(hipGetLastError(), "conv0_wgrad launch");
  if (rc) return rc;
  hipLaunchKernelGGL(conv0_wgrad_reduce_kernel, dim3(9 * Ci + 1), dim3(256), 0, s, slab, bias_slab, dw, dbias,
                     blocks, 9 * Ci, Co);
  return check_hip(hipGetLastError(), "conv0_wgrad_reduce launch");
}

int vqa_conv0_wgrad_bf16(const void* x_nchw, int x_is_fp16, const void* dpooled_bf16, const uint8_t* argmax, float* dw,
                         float* dbias, int B, int Ci, int H, int W, int Co, float* workspace, int64_t workspace_bytes,
                         vqa_stream_t stream) {
  VQA_REQUIRE(x_nchw && dpooled_bf16 && argmax && dw && dbias && workspace, "vqa_conv0_wgrad_bf16: null pointer");
  VQA_REQUIRE(c0_supported(Ci, H, W, Co, 1) && Ci <= 3, "vqa_conv0_wgrad_bf16: unsupported shape Ci=%d H=%d W=%d Co=%d", Ci, H, W, Co);
  VQA_REQUIRE(((uintptr_t)dpooled_bf16 % 16) == 0 && ((uintptr_t)argmax % 8) == 0, "vqa_conv0_wgrad_bf16: dpooled / argmax alignment");
  if (workspace_bytes < vqa_conv0_wgrad_workspace_bytes(Co)) {
    set_error("vqa_conv0_wgrad_bf16: workspace too small");
    return VQA_ERR_WORKSPACE;
  }
  const int Hp = (H - 2) / 2, Wp = (W - 2) / 2;
  const int NG = (2 * Wp + 15) / 16;
  const int RSTR = 32 * NG + 16;                       // bytes per staged image row (+16: rows start on different banks)
  size_t lds = (size_t)3 * Ci * 4 * RSTR + (size_t)8 * NG * Co * 3;
  if (lds < (size_t)(4 * 32 + 4) * Co * 4) lds = (size_t)(4 * 32 + 4) * Co * 4;   // the end-of-kernel combine area
  lds = (lds + 15) & ~(size_t)15;
  VQA_REQUIRE(lds <= 160 * 1024 && W <= 512 && 8 * NG * Co <= 16384, "vqa_conv0_wgrad_bf16: image too wide (W=%d)", W);
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 2) per_cu = 2;                          // 2 waves per SIMD: the prefetch registers
  int blocks = 256 * per_cu;
  if (blocks > B * Hp) blocks = B * Hp;
  float* slab = workspace;
  float* bias_slab = workspace + (int64_t)kC0Blocks * 32 * Co;
  hipStream_t s = (hipStream_t)stream;
  C0_DISPATCH(Ci, Co / 32, {
    if constexpr (kCI <= 3) {
      if (x_is_fp16) {
        auto kern = conv0_wgrad_bf16_kernel<kCI, kTN, true>;
        int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)lds, "attr(conv0_wgrad_bf16)");
        if (rc0) return rc0;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, x_nchw, static_cast<const uint16_t*>(dpooled_bf16), argmax, slab,
                           bias_slab, B, H, W, Hp, Wp, RSTR, NG);
      } else {
        auto kern = conv0_wgrad_bf16_kernel<kCI, kTN, false>;
        int rc0 = ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)lds, "attr(conv0_wgrad_bf16)");
        if (rc0) return rc0;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, x_nchw, static_cast<const uint16_t*>(dpooled_bf16), argmax, slab,
                           bias_slab, B, H, W, Hp, Wp, RSTR, NG);
      }
    }
  });
  int rc = check_hip(hipGetLastError(), "conv0_wgrad_bf16 launch");
  if (rc) return rc;
  hipLaunchKernelGGL(conv0_wgrad_reduce_kernel, dim3(9 * Ci + 1), dim3(256), 0, s, slab, bias_slab, dw, dbias,
                     blocks, 9 * Ci, Co);
  return check_hip(hipGetLastError(), "conv0_wgrad_reduce launch");
}

}  // extern "C"
