// vqa_gemm in fp32 on the bf16 matrix cores (x3_core.hpp: exact three-way bf16 operand split, six partial products,
// fp32 accumulate): same operands, layouts, epilogue and split-K reduction as gemm.hip -- only the K loop differs.
// For the large contractions of the attention stage (v_conv forward / dW / dX, models/model.py:173,187-193).
#include "x3_core.hpp"
#include "gemm_epilogue.hpp"

// Compiled five times (dl_vqa_amd/build.py): VQA_GEMM_PART = 0 is the host side + the C ABI, parts 1-4 hold the kernels of
// one operand layout each (the fused epilogue's straight-line variants are slow to compile).
#ifndef VQA_GEMM_PART
#define VQA_GEMM_PART 0
#endif

namespace vqa {

#ifndef VQA_X3_PF
#define VQA_X3_PF 2
#endif
using CfgG = TileCfg<192, 128, 2, 2, 4, VQA_X3_PF>;   // one workgroup per CU: 4 MFMA waves of 96 x 64 + 4 loader waves

template <class Cfg, class AL, class BL>
__global__ __launch_bounds__(Cfg::THREADS, 2) void gemm_x3_kernel(typename AL::Params pa, typename BL::Params pb, EpiParams pe,
                                                                  int tiles_m, int tiles_n, int nk, int ks_per_split,
                                                                  int splits) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const TileCoord tc = tile_coord(tiles_m, tiles_n, 0, splits);
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
  if (pe.slab) {
    store_acc_tiles<Cfg>(acc, pe.slab + (int64_t)tc.split * pe.M * pe.N, pe.N, pe.M, pe.N, m0, n0, wm, wn, lane);
    return;
  }
  gemm_epilogue<Cfg>(pe, acc, m0, n0, wm, wn, lane);
}

// persistent variant (no split-K): 256 workgroups walk the tiles, the loaders run ahead into the next tile during the
// epilogue -- with K = 256 (v_conv forward: 8 K-steps per tile) prologue and epilogue are most of a tile's life
template <class Cfg, class AL, class BL>
__global__ __launch_bounds__(Cfg::THREADS, 2) void gemm_x3_persistent_kernel(typename AL::Params pa, typename BL::Params pb,
                                                                             EpiParams pe, int tiles_m, int tiles_n, int nk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  gemm_persistent_x<Cfg, AL, BL>(
      xcd_swizzle(blockIdx.x, gridDim.x), gridDim.x, tiles_m * tiles_n, nk, smem,
      [&](int t, AL& al, BL& bl) {
        const int mt = t / tiles_n, nt = t - mt * tiles_n;
        al.init(pa, mt * Cfg::BM, loader_tid<Cfg>(), 0);
        bl.init(pb, nt * Cfg::BN, loader_tid<Cfg>(), 0);
      },
      [&](int t, f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
        const int mt = t / tiles_n, nt = t - mt * tiles_n;
        gemm_epilogue<Cfg>(pe, acc, mt * Cfg::BM, nt * Cfg::BN, wm, wn, lane);
      });
}

struct GemmPlanX { int tiles_m, tiles_n, nk, splits, ks_per_split; };
static GemmPlanX plan_gemm_x3(int M, int N, int K) {
  GemmPlanX p;
  p.tiles_m = (M + CfgG::BM - 1) / CfgG::BM;
  p.tiles_n = (N + CfgG::BN - 1) / CfgG::BN;
  p.nk = (K + BK - 1) / BK;
  const int tiles = p.tiles_m * p.tiles_n;
  int splits = 1;
  if (tiles < 128) {                       // one workgroup per CU: fill the 256 slots along K
    splits = 256 / tiles;
    const int max_splits = p.nk / 8 > 1 ? p.nk / 8 : 1;
    if (splits > max_splits) splits = max_splits;
    if (splits > 64) splits = 64;
  }
  p.ks_per_split = (p.nk + splits - 1) / splits;
  p.splits = (p.nk + p.ks_per_split - 1) / p.ks_per_split;
  return p;
}

template <class AL, class BL>
int launch_gemm_x3(const typename AL::Params& pa, const typename BL::Params& pb, const EpiParams& pe,
                          const GemmPlanX& p, hipStream_t s) {
  using SL = SmemLayoutX<CfgG, AL::kTypeR, BL::kTypeR>;
  if (p.splits == 1 && p.tiles_m * p.tiles_n > 256 && knobs().persistent != 0) {
    auto pk = gemm_x3_persistent_kernel<CfgG, AL, BL>;
    int rc = ensure_dyn_smem(reinterpret_cast<const void*>(pk), SL::BYTES, "hipFuncSetAttribute(gemm_x3_p)");
    if (rc) return rc;
    hipLaunchKernelGGL(pk, dim3(256), dim3(CfgG::THREADS), SL::BYTES, s, pa, pb, pe, p.tiles_m, p.tiles_n, p.nk);
    return check_hip(hipGetLastError(), "gemm_x3_persistent_kernel launch");
  }
  auto kern = gemm_x3_kernel<CfgG, AL, BL>;
  int rc = ensure_dyn_smem(reinterpret_cast<const void*>(kern), SL::BYTES, "hipFuncSetAttribute(gemm_x3)");
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n * p.splits), dim3(CfgG::THREADS), SL::BYTES, s, pa, pb, pe, p.tiles_m,
                     p.tiles_n, p.nk, p.ks_per_split, p.splits);
  return check_hip(hipGetLastError(), "gemm_x3_kernel launch");
}

#define VQA_GX3_LAUNCH(KW, AL, BL)                                                                                 \
  KW template int launch_gemm_x3<AL<CfgG::NVA, CfgG::LT>, BL<CfgG::NVB, CfgG::LT>>(                                  \
      const typename AL<CfgG::NVA, CfgG::LT>::Params&, const typename BL<CfgG::NVB, CfgG::LT>::Params&, const EpiParams&, \
      const GemmPlanX&, hipStream_t);
#if VQA_GEMM_PART == 1
VQA_GX3_LAUNCH(, PlainR, PlainR)
#else
VQA_GX3_LAUNCH(extern, PlainR, PlainR)
#endif
#if VQA_GEMM_PART == 2
VQA_GX3_LAUNCH(, PlainR, PlainC)
#else
VQA_GX3_LAUNCH(extern, PlainR, PlainC)
#endif
#if VQA_GEMM_PART == 3
VQA_GX3_LAUNCH(, PlainC, PlainR)
#else
VQA_GX3_LAUNCH(extern, PlainC, PlainR)
#endif
#if VQA_GEMM_PART == 4
VQA_GX3_LAUNCH(, PlainC, PlainC)
#else
VQA_GX3_LAUNCH(extern, PlainC, PlainC)
#endif

}  // namespace vqa

#if VQA_GEMM_PART == 0
using namespace vqa;

extern "C" {

int64_t vqa_gemm_x3_workspace_bytes(int M, int N, int K) {
  const GemmPlanX p = plan_gemm_x3(M, N, K);
  return p.splits > 1 ? (int64_t)p.splits * M * N * 4 : 0;
}

int vqa_gemm_x3(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C, int64_t ldc,
                int M, int N, int K, const float* bias1, const float* bias2, const float* rowgroup, int64_t rg_ld,
                int rg_div, int rg_op, int relu, int accumulate, float* aux, float* workspace, int64_t workspace_bytes,
                int tag, vqa_stream_t stream) {
  VQA_REQUIRE(A && B && C, "vqa_gemm_x3: null operand");
  VQA_REQUIRE(M > 0 && N > 0 && K > 0, "vqa_gemm_x3: bad shape M=%d N=%d K=%d", M, N, K);
  VQA_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && lda % 4 == 0 && ldb % 4 == 0,
              "vqa_gemm_x3: A/B must be 16-byte aligned with leading dimensions multiple of 4 (lda=%lld ldb=%lld)",
              (long long)lda, (long long)ldb);
  VQA_REQUIRE(lda < (1 << 21) && ldb < (1 << 21) && ldc < (1 << 21),
              "vqa_gemm_x3: leading dimensions must be below 2^21 (lda=%lld ldb=%lld ldc=%lld)", (long long)lda,
              (long long)ldb, (long long)ldc);
  VQA_REQUIRE(!rowgroup || rg_div > 0, "vqa_gemm_x3: rg_div must be positive");
  hipStream_t s = (hipStream_t)stream;
  const GemmPlanX p = plan_gemm_x3(M, N, K);
  EpiParams pe{C, ldc, M, N, bias1, bias2, rowgroup, rg_ld, rg_div, rg_op, relu, accumulate, aux, nullptr, nullptr};
  if (p.splits > 1) {
    const int64_t need = (int64_t)p.splits * M * N * 4;
    if (!workspace || workspace_bytes < need) {
      set_error("vqa_gemm_x3: workspace %lld bytes < %lld needed", (long long)workspace_bytes, (long long)need);
      return VQA_ERR_WORKSPACE;
    }
    pe.slab = workspace;
  }
  set_launch_tag(tag);
  ProfScope prof(VQA_K_GEMM, s);
  using AR = PlainR<CfgG::NVA, CfgG::LT>; using AC = PlainC<CfgG::NVA, CfgG::LT>;
  using BR = PlainR<CfgG::NVB, CfgG::LT>; using BC = PlainC<CfgG::NVB, CfgG::LT>;
  int rc;
  if (!transA && transB) rc = launch_gemm_x3<AR, BR>({A, lda, M, K}, {B, ldb, N, K}, pe, p, s);
  else if (!transA && !transB) rc = launch_gemm_x3<AR, BC>({A, lda, M, K}, {B, ldb, N, K}, pe, p, s);
  else if (transA && transB) rc = launch_gemm_x3<AC, BR>({A, lda, M, K}, {B, ldb, N, K}, pe, p, s);
  else rc = launch_gemm_x3<AC, BC>({A, lda, M, K}, {B, ldb, N, K}, pe, p, s);
  if (rc) return rc;
  if (p.splits > 1) rc = launch_splitk_reduce(pe, p.splits, s);
  return rc;
}

}  // extern "C"
#endif  // VQA_GEMM_PART == 0
