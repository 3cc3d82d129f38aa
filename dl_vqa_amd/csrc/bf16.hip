// bf16 path (BASELINE configs[3]): plain GEMMs on v_mfma_f32_32x32x16_bf16 and the fp32 <-> bf16 converters.
//
// Reference ops re-typed here: nn.Linear / 1x1 nn.Conv2d contractions (models/model.py:173-174,178,202,205) with bf16
// operands and fp32 accumulation; parameters stay fp32 (master copy, Adam), a bf16 copy is made per step.
// The convolution kernels of the bf16 path live in conv.hip (conv_bf16.inc).
#include "bf16_core.hpp"
#include "gemm_epilogue.hpp"
#include <stdlib.h>

namespace vqa {

using CfgB128 = TileCfg<128, 128, 2, 2>;
using CfgB64 = TileCfg<64, 64, 2, 2>;
// 128 x 256: a weight-gradient product whose output is only 256 columns wide (v_conv dW: [mid] x [C = 256], K = B * positions) --
// with two 128-column tiles the reduction-major A operand (dx', 3 GB) was fetched 1.7x (PMC); one tile spans all columns
using CfgB128x256 = TileCfg<128, 256, 2, 2>;

template <class Cfg, class AL, class BL>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::MIN_WAVES) void gemm_bf16_kernel(typename AL::Params pa,
                                                                              typename BL::Params pb, EpiParams pe,
                                                                              int tiles_m, int tiles_n, int nk,
                                                                              int ks_per_split, int order, int splits) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const TileCoord tc = tile_coord(tiles_m, tiles_n, order, splits);
  const int m0 = tc.mt * Cfg::BM, n0 = tc.nt * Cfg::BN;
  const int ks0 = tc.split * ks_per_split;
  const int ks1 = min(nk, ks0 + ks_per_split);
  f32x16 acc[Cfg::TM][Cfg::TN];
  acc_zero<Cfg>(acc);
  if (!gemm_mainloop_b<Cfg, AL, BL>(
          [&](AL& al, BL& bl) {
            al.init(pa, m0, loader_tid<Cfg>(), ks0);
            bl.init(pb, n0, loader_tid<Cfg>(), ks0);
          },
          acc, ks0, ks1, smem))
    return;
  float* slab = pe.slab ? pe.slab + (int64_t)tc.split * pe.M * pe.N : nullptr;
  if (slab) {
    store_acc_tiles<Cfg>(acc, slab, pe.N, pe.M, pe.N, m0, n0, wm, wn, lane);
    return;
  }
  if constexpr (Cfg::WN == 64) {
    // bf16 output of an interior tile, no aux / accumulate, row groups at least a tile tall: 16-byte stores through LDS
    const bool staged = pe.Cb != nullptr && !pe.aux && !pe.accumulate && (!pe.rg || pe.rg_div >= Cfg::BM) &&
                        m0 + Cfg::BM <= pe.M && n0 + Cfg::BN <= pe.N && (pe.ldc & 7) == 0 &&
                        (reinterpret_cast<uintptr_t>(pe.Cb) & 15) == 0;
    if (staged) return gemm_epilogue_bf16_staged<Cfg>(pe, acc, m0, n0, wm, wn, lane, smem);
  }
  gemm_epilogue<Cfg>(pe, acc, m0, n0, wm, wn, lane);
}

template <class Cfg, class AL, class BL>
static int launch_gemm_bf16(const typename AL::Params& pa, const typename BL::Params& pb, const EpiParams& pe,
                            const GemmPlan& p, hipStream_t s) {
  using SL = SmemLayoutB<Cfg, AL::kTypeR, BL::kTypeR>;
  auto kern = gemm_bf16_kernel<Cfg, AL, BL>;
  int rc = ensure_dyn_smem(reinterpret_cast<const void*>(kern), SL::BYTES, "hipFuncSetAttribute(gemm_bf16)");
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n * p.splits), dim3(Cfg::THREADS), SL::BYTES, s, pa, pb, pe,
                     p.tiles_m, p.tiles_n, p.nk, p.ks_per_split, p.order, p.splits);
  return check_hip(hipGetLastError(), "gemm_bf16_kernel launch");
}

// A: transA = 0 -> [M][K] (type R), 1 -> [K][M] (type C);  B: transB = 1 -> [N][K] (type R), 0 -> [K][N] (type C)
template <class Cfg>
static int dispatch_gemm_bf16(const void* A, int64_t lda, int transA, const void* B, int64_t ldb, int transB,
                              const EpiParams& pe, const GemmPlan& p, int M, int N, int K, hipStream_t s) {
  using AR = PlainR<Cfg::NVA, Cfg::LT>; using AC = PlainCb<Cfg::BM, Cfg::LT>;
  using BR = PlainR<Cfg::NVB, Cfg::LT>; using BC = PlainCb<Cfg::BN, Cfg::LT>;
  const float* Af = static_cast<const float*>(A);
  const float* Bf = static_cast<const float*>(B);
  if (!transA && transB) return launch_gemm_bf16<Cfg, AR, BR>({Af, lda / 2, M, K / 2}, {Bf, ldb / 2, N, K / 2}, pe, p, s);
  if (!transA && !transB) return launch_gemm_bf16<Cfg, AR, BC>({Af, lda / 2, M, K / 2}, {B, ldb, N, K}, pe, p, s);
  if (transA && transB) return launch_gemm_bf16<Cfg, AC, BR>({A, lda, M, K}, {Bf, ldb / 2, N, K / 2}, pe, p, s);
  return launch_gemm_bf16<Cfg, AC, BC>({A, lda, M, K}, {B, ldb, N, K}, pe, p, s);
}

// ------------------------------------------------------------------ converters
// y[i] = bf16(x[i]); 8 values per thread and iteration (32 bytes in, 16 bytes out)
__global__ void f32_to_bf16_kernel(const float* x, uint16_t* y, int64_t n) {
  const int64_t n8 = n / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
    uint4 o;
    o.x = pack_bf16x2(a.x, a.y); o.y = pack_bf16x2(a.z, a.w); o.z = pack_bf16x2(b.x, b.y); o.w = pack_bf16x2(b.z, b.w);
    reinterpret_cast<uint4*>(y)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < n - 8 * n8) y[8 * n8 + threadIdx.x] = bf16_bits(x[8 * n8 + threadIdx.x]);
}
// y = bf16(x * keep(seed, i) / (1 - p)); same counter hash and index convention as vqa_dropout
__global__ void dropout_to_bf16_kernel(const float* x, uint16_t* y, int64_t n, float p, float inv_keep, uint64_t seed) {
  const int64_t n8 = n / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    if (p > 0.f) {
#pragma unroll
      for (int k = 0; k < 8; k += 4) {
        const float4 ds_ = drop_scale4(seed, (uint64_t)(8 * i + k), p, inv_keep);
        v[k] *= ds_.x; v[k + 1] *= ds_.y; v[k + 2] *= ds_.z; v[k + 3] *= ds_.w;
      }
    }
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    reinterpret_cast<uint4*>(y)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < n - 8 * n8) {
    const int64_t i = 8 * n8 + threadIdx.x;
    y[i] = bf16_bits(x[i] * (p > 0.f ? drop_scale(seed, (uint64_t)i, p, inv_keep) : 1.f));
  }
}
__global__ void bf16_to_f32_kernel(const uint16_t* x, float* y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = __uint_as_float((uint32_t)x[i] << 16);
}
// y[c][r] = bf16(x[r][c]) for a row-major [rows][cols] matrix: the bf16 copy of a weight in the orientation that makes
// it a k-contiguous (type R) operand of the backward-data GEMM.  32 x 32 tiles through LDS.
__global__ void f32_to_bf16_transpose_kernel(const float* x, uint16_t* y, int rows, int cols) {
  __shared__ float t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  for (int k = ty; k < 32; k += 8)
    if (r0 + k < rows && c0 + tx < cols) t[k][tx] = x[(int64_t)(r0 + k) * cols + c0 + tx];
  __syncthreads();
  for (int k = ty; k < 32; k += 8)
    if (c0 + k < cols && r0 + tx < rows) y[(int64_t)(c0 + k) * rows + r0 + tx] = bf16_bits(t[tx][k]);
}

}  // namespace vqa

using namespace vqa;

extern "C" {

int64_t vqa_gemm_bf16_workspace_bytes(int M, int N, int K) {
  const GemmPlan p = plan_gemm(M, N, K, BKB);
  return p.splits > 1 ? (int64_t)p.splits * M * N * 4 : 0;
}

int vqa_gemm_bf16(const void* A, int64_t lda, int transA, const void* B, int64_t ldb, int transB, void* C,
                  int64_t ldc, int c_is_bf16, int M, int N, int K, const float* bias1, const float* bias2,
                  const float* rowgroup, int64_t rg_ld, int rg_div, int rg_op, int relu, int accumulate, float* aux,
                  float* workspace, int64_t workspace_bytes, int tag, vqa_stream_t stream) {
  VQA_REQUIRE(A && B && C, "vqa_gemm_bf16: null operand");
  VQA_REQUIRE(M > 0 && N > 0 && K > 0, "vqa_gemm_bf16: bad shape M=%d N=%d K=%d", M, N, K);
  VQA_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && lda % 8 == 0 && ldb % 8 == 0 && K % 8 == 0,
              "vqa_gemm_bf16: A/B must be 16-byte aligned, leading dimensions and K multiples of 8 (lda=%lld ldb=%lld K=%d)",
              (long long)lda, (long long)ldb, K);
  VQA_REQUIRE((!transA || M % 8 == 0) && (transB || N % 8 == 0),
              "vqa_gemm_bf16: a reduction-major operand needs its row length to be a multiple of 8 (M=%d N=%d)", M, N);
  VQA_REQUIRE(lda < (1 << 21) && ldb < (1 << 21) && ldc < (1 << 21), "vqa_gemm_bf16: leading dimensions must be below 2^21");
  VQA_REQUIRE(!rowgroup || rg_div > 0, "vqa_gemm_bf16: rg_div must be positive");
  VQA_REQUIRE(!(c_is_bf16 && accumulate), "vqa_gemm_bf16: accumulate needs an fp32 C");
  hipStream_t s = (hipStream_t)stream;
  const GemmPlan p = plan_gemm(M, N, K, BKB);
  EpiParams pe{c_is_bf16 ? nullptr : static_cast<float*>(C), ldc, M, N, bias1, bias2, rowgroup, rg_ld, rg_div, rg_op,
               relu, accumulate, aux, nullptr, c_is_bf16 ? static_cast<uint16_t*>(C) : nullptr};
  if (p.splits > 1) {
    const int64_t need = (int64_t)p.splits * M * N * 4;
    if (!workspace || workspace_bytes < need) {
      set_error("vqa_gemm_bf16: workspace %lld bytes < %lld needed", (long long)workspace_bytes, (long long)need);
      return VQA_ERR_WORKSPACE;
    }
    pe.slab = workspace;
  }
  set_launch_tag(tag);
  ProfScope prof(VQA_K_GEMM, s);
  static const int wide_ok = [] { const char* e = getenv("VQA_GEMM_WIDE"); return e && atoi(e) == 0 ? 0 : 1; }();
  if (wide_ok && transA && !transB && N == 256 && M % 128 == 0 && K >= (1 << 16) && workspace) {
    // long-K weight gradient with a 256-column output: 128 x 256 tiles, one workgroup per CU, split-K over the 256 slots
    GemmPlan q = p;
    q.big = 1;
    q.tiles_m = M / 128;
    q.tiles_n = 1;
    int splits = 256 / q.tiles_m;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    q.ks_per_split = (q.nk + splits - 1) / splits;
    q.splits = (q.nk + q.ks_per_split - 1) / q.ks_per_split;
    q.order = 0;
    if (workspace_bytes >= (int64_t)q.splits * M * N * 4) {
      pe.slab = q.splits > 1 ? workspace : nullptr;
      using Cfg = CfgB128x256;
      int rc = launch_gemm_bf16<Cfg, PlainCb<Cfg::BM, Cfg::LT>, PlainCb<Cfg::BN, Cfg::LT>>({A, lda, M, K}, {B, ldb, N, K}, pe, q, s);
      if (rc) return rc;
      if (q.splits > 1) rc = launch_splitk_reduce(pe, q.splits, s);
      return rc;
    }
  }
  int rc = p.big ? dispatch_gemm_bf16<CfgB128>(A, lda, transA, B, ldb, transB, pe, p, M, N, K, s)
                 : dispatch_gemm_bf16<CfgB64>(A, lda, transA, B, ldb, transB, pe, p, M, N, K, s);
  if (rc) return rc;
  if (p.splits > 1) rc = launch_splitk_reduce(pe, p.splits, s);
  return rc;
}

int vqa_f32_to_bf16(const float* x, void* y_bf16, int64_t n, vqa_stream_t stream) {
  VQA_REQUIRE(x && y_bf16 && n > 0, "vqa_f32_to_bf16: bad args");
  VQA_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y_bf16 % 16) == 0, "vqa_f32_to_bf16: pointers must be 16-byte aligned");
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                     static_cast<uint16_t*>(y_bf16), n);
  return check_hip(hipGetLastError(), "f32_to_bf16 launch");
}

int vqa_dropout_to_bf16(const float* x, void* y_bf16, int64_t n, float p, uint64_t seed, vqa_stream_t stream) {
  VQA_REQUIRE(x && y_bf16 && n > 0 && p >= 0.f && p < 1.f, "vqa_dropout_to_bf16: bad args");
  VQA_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y_bf16 % 16) == 0, "vqa_dropout_to_bf16: pointers must be 16-byte aligned");
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dropout_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                     static_cast<uint16_t*>(y_bf16), n, p, p > 0.f ? 1.0f / (1.0f - p) : 1.0f, seed);
  return check_hip(hipGetLastError(), "dropout_to_bf16 launch");
}

int vqa_bf16_to_f32(const void* x_bf16, float* y, int64_t n, vqa_stream_t stream) {
  VQA_REQUIRE(x_bf16 && y && n > 0, "vqa_bf16_to_f32: bad args");
  int64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const uint16_t*>(x_bf16), y, n);
  return check_hip(hipGetLastError(), "bf16_to_f32 launch");
}

int vqa_f32_to_bf16_transpose(const float* x, void* y_bf16, int rows, int cols, vqa_stream_t stream) {
  VQA_REQUIRE(x && y_bf16 && rows > 0 && cols > 0, "vqa_f32_to_bf16_transpose: bad args");
  hipLaunchKernelGGL(f32_to_bf16_transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0,
                     (hipStream_t)stream, x, static_cast<uint16_t*>(y_bf16), rows, cols);
  return check_hip(hipGetLastError(), "f32_to_bf16_transpose launch");
}

}  // extern "C"
