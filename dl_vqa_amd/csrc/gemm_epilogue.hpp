// Fused GEMM epilogue + split-K reduction, shared by the fp32 engine (gemm.hip) and the bf16 engine (bf16.hip).
#pragma once
#include <type_traits>
#include "gemm_core.hpp"

namespace vqa {

// ------------------------------------------------------------------ epilogue
struct EpiParams {
  float* C; int64_t ldc; int M, N;
  const float* bias1; const float* bias2;
  const float* rg; int64_t rg_ld; int rg_div; int rg_op;
  int relu; int accumulate;
  float* aux;   // optional: raw product before the epilogue, same shape/ld as C
  float* slab;  // != nullptr: split-K partials [split][M][N]
  uint16_t* Cb; // != nullptr: the result is stored as bf16 here (same ldc, in elements) instead of fp32 in C
#ifdef VQA_DIAG
  unsigned long long* bar_dbg;   // diagnostic build: [0] barriers executed by loader waves, [1] loader waves, [2] / [3] the same for MFMA waves
#endif
#ifdef VQA_EXP_EPI_DROPOUT
  // Reconstruction of the round-2 experiment that was withdrawn after a hang (tools/build_variant.sh hangrepro
  // -DVQA_EXP_EPI_DROPOUT): dropout of the attention branch inside the GEMM epilogue.  Never part of the shipped library.
  float drop_p; float drop_inv; uint64_t drop_seed;
#endif
};

__device__ __forceinline__ uint16_t epi_bf16(float x) {       // round to nearest even (v_cvt_pk_bf16_f32)
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 v = {(__bf16)x, (__bf16)0.f};
  return (uint16_t)(__builtin_bit_cast(uint32_t, v) & 0xffffu);
}

__device__ __forceinline__ float epi_apply(const EpiParams& e, float v, int row, int col) {
  if (e.rg) {
    const float g = e.rg[(int64_t)(row / e.rg_div) * e.rg_ld + col];
    v = e.rg_op ? v * g : v + g;
  }
  if (e.bias1) v += e.bias1[col];
  if (e.bias2) v += e.bias2[col];
  if (e.relu) v = fmaxf(v, 0.f);
  if (e.accumulate) v += e.C[(int64_t)row * e.ldc + col];
  return v;
}

// Fused epilogue.  gfx950 counts stores and loads in the same vmcnt, so a load anywhere inside the per-element
// code makes every element wait for the previous element's STORE to be acknowledged (measured: 64 elements x
// ~1000 cycles per tile, more than the K loop of a K = 256 GEMM).  The per-element code is therefore
// load-free: the mode is decided once per tile, per-column terms and the (at most two) row-group terms of a
// tile are loaded up front, and the `accumulate` / general row-group loads of a 32x32 accumulator tile are
// issued as one batch of 16 before its 16 stores.
//   RG: 0 none, 1 row-group term with rg_div >= BM (the tile spans at most two groups: q' tiled over the
//       676 image positions of a sample), 2 general (one division per element)
// Addressing: a 32x32 accumulator tile is stored through a buffer resource whose base is the tile's first
// element (scalar); the lane's part -- row 4*(lane>>5), column lane&31 -- is one VGPR for the whole epilogue
// and element r's row offset is a scalar, so a store costs no VALU instruction.  Interior tiles (the common
// case) carry no per-element predicate either; on edge tiles an invalid element's offset becomes BUF_OOB
// and the hardware drops the store.
template <class Cfg, int RG, bool ACC, bool AUX>
__device__ __forceinline__ void gemm_epilogue_mode(const EpiParams& pe, f32x16 (&acc)[Cfg::TM][Cfg::TN], int m0,
                                                   int n0, int wm, int wn, int lane) {
  const int g0 = RG ? m0 / pe.rg_div : 0;
  const int boundary = (g0 + 1) * pe.rg_div;
  const bool mul = pe.rg_op != 0, relu = pe.relu != 0;
  // ACC / AUX say what the instantiation supports; the general instantiations still honour the run-time flags
  const bool accum = ACC && pe.accumulate != 0, aux = AUX && pe.aux != nullptr;
  const bool interior = m0 + Cfg::BM <= pe.M && n0 + Cfg::BN <= pe.N;       // uniform
  const bool one_group = RG == 1 && boundary >= m0 + Cfg::BM;               // uniform: rg1 never selected
  const uint32_t ldb = (uint32_t)pe.ldc * 4u;                               // ldc < 2^21 (checked on entry)
  const uint32_t vlane = (uint32_t)(4 * (lane >> 5)) * ldb + 4u * (uint32_t)(lane & 31);
  const bool ob = pe.Cb != nullptr;                                         // uniform: bf16 result
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const int colt = n0 + wn * Cfg::WN + 32 * j;          // tile's first column (uniform)
    const int col = colt + (lane & 31);
    const bool cok = col < pe.N;
    const int cc = cok ? col : 0;
    float cb = 0.f;
    if (pe.bias1) cb += pe.bias1[cc];
    if (pe.bias2) cb += pe.bias2[cc];
    float rg0 = 0.f, rg1 = 0.f;
    if (RG == 1) {
      rg0 = pe.rg[(int64_t)g0 * pe.rg_ld + cc];
      rg1 = (!one_group && boundary < pe.M) ? pe.rg[(int64_t)(g0 + 1) * pe.rg_ld + cc] : rg0;
    }
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) {
      const int rowt = m0 + wm * Cfg::WM + 32 * i;        // tile's first row (uniform)
      const int row0 = rowt + 4 * (lane >> 5);
      const __amdgpu_buffer_rsrc_t rc = buf_rsrc(pe.C + (int64_t)rowt * pe.ldc + colt);
      const __amdgpu_buffer_rsrc_t rx = buf_rsrc((aux ? pe.aux : pe.C) + (int64_t)rowt * pe.ldc + colt);
      const __amdgpu_buffer_rsrc_t rb = buf_rsrc((ob ? pe.Cb : reinterpret_cast<uint16_t*>(pe.C)) + (int64_t)rowt * pe.ldc + colt);
      auto tile = [&](auto inner) {
        constexpr bool INNER = decltype(inner)::value;
        auto vo = [&](int dr) { return INNER || (cok && row0 + dr < pe.M) ? vlane : BUF_OOB; };
        float old[16], g[16];
        if (accum) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            old[r] = __uint_as_float(buf_load4(rc, vo(dr), (uint32_t)dr * ldb));
          }
        }
        if (RG == 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2);
            g[r] = pe.rg[(int64_t)((row < pe.M ? row : 0) / pe.rg_div) * pe.rg_ld + cc];
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          float v = acc[i][j][r];
          if (aux) buf_store4(rx, v, vo(dr), (uint32_t)dr * ldb);
          if (RG) {
            const float t = RG == 2 ? g[r] : (one_group || row0 + dr < boundary) ? rg0 : rg1;
            v = mul ? v * t : v + t;
          }
          v += cb;
          if (relu) v = fmaxf(v, 0.f);
          if (accum) v += old[r];
#ifdef VQA_EXP_EPI_DROPOUT
          if (pe.drop_p > 0.f)
            v *= drop_scale(pe.drop_seed, (uint64_t)(row0 + dr) * (uint64_t)pe.N + (uint64_t)col, pe.drop_p, pe.drop_inv);
#endif
          if (ob) {
            const uint32_t o = vo(dr);
            __builtin_amdgcn_raw_buffer_store_b16(epi_bf16(v), rb, (int)(o == BUF_OOB ? BUF_OOB : o >> 1),
                                                  (int)((uint32_t)dr * (ldb >> 1)), 0);
          } else {
#ifdef VQA_EXP_NOSTORE_ALL   // timing experiments only: the epilogue's arithmetic without its store instructions
            if (pe.M < 0)
#endif
            buf_store4(rc, v, vo(dr), (uint32_t)dr * ldb);
          }
        }
      };
      if (interior) tile(std::true_type{}); else tile(std::false_type{});
    }
  }
}

template <class Cfg>
__device__ __forceinline__ void gemm_epilogue(const EpiParams& pe, f32x16 (&acc)[Cfg::TM][Cfg::TN], int m0, int n0,
                                              int wm, int wn, int lane) {
  const int rg = !pe.rg ? 0 : (pe.rg_div >= Cfg::BM ? 1 : 2);
  const bool a = pe.accumulate != 0, x = pe.aux != nullptr;
  // the combinations the train step uses get their own straight-line code; the rest share the general one
  if (rg == 0 && !a && !x) return gemm_epilogue_mode<Cfg, 0, false, false>(pe, acc, m0, n0, wm, wn, lane);
  if (rg == 0 && a && !x) return gemm_epilogue_mode<Cfg, 0, true, false>(pe, acc, m0, n0, wm, wn, lane);
  if (rg == 1 && !a && !x) return gemm_epilogue_mode<Cfg, 1, false, false>(pe, acc, m0, n0, wm, wn, lane);
  if (rg == 0 && !a && x) return gemm_epilogue_mode<Cfg, 0, false, true>(pe, acc, m0, n0, wm, wn, lane);
  if (rg == 1) return gemm_epilogue_mode<Cfg, 1, true, true>(pe, acc, m0, n0, wm, wn, lane);
  if (rg == 2) return gemm_epilogue_mode<Cfg, 2, true, true>(pe, acc, m0, n0, wm, wn, lane);
  return gemm_epilogue_mode<Cfg, 0, true, true>(pe, acc, m0, n0, wm, wn, lane);
}

// bf16 result through LDS: the direct form above issues one 2-byte store per accumulator register (64 store instructions of
// 128 bytes per wave and tile): for the v_conv forward of the bf16 path -- 3 GB of x = relu(v' + q') per step -- that was
// store-ISSUE bound at ~1.1 TB/s.  Here an MFMA wave writes its WM x WN block as bf16 into a wave-private LDS scratch
// (the stage buffers are free: the last K-step barrier has passed and the loader waves have exited) and stores it with
// 16 bytes per lane, whole 128-byte runs of a row.  Interior tiles without aux / accumulate only (uniform test by the caller).
template <class Cfg>
__device__ __forceinline__ void gemm_epilogue_bf16_staged(const EpiParams& pe, f32x16 (&acc)[Cfg::TM][Cfg::TN], int m0, int n0,
                                                          int wm, int wn, int lane, float* smem) {
  static_assert(Cfg::WN == 64, "64-column wave blocks: a row of the scratch is one 128-byte run");
  constexpr int PITCH = 128;                                           // bytes per scratch row (64 bf16)
  char* const scr = reinterpret_cast<char*>(smem) + (wm * Cfg::WAVES_N + wn) * (Cfg::WM * PITCH);
  const int RG = !pe.rg ? 0 : 1;
  const int g0 = RG ? m0 / pe.rg_div : 0;
  const int boundary = (g0 + 1) * pe.rg_div;
  const bool mul = pe.rg_op != 0, relu = pe.relu != 0;
  const bool one_group = RG == 1 && boundary >= m0 + Cfg::BM;
  const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const int col = n0 + wn * Cfg::WN + 32 * j + l31;
    float cb = 0.f;
    if (pe.bias1) cb += pe.bias1[col];
    if (pe.bias2) cb += pe.bias2[col];
    float rg0 = 0.f, rg1 = 0.f;
    if (RG == 1) {
      rg0 = pe.rg[(int64_t)g0 * pe.rg_ld + col];
      rg1 = (!one_group && boundary < pe.M) ? pe.rg[(int64_t)(g0 + 1) * pe.rg_ld + col] : rg0;
    }
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) {
      const int row0 = m0 + wm * Cfg::WM + 32 * i + 4 * h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        float v = acc[i][j][r];
        if (RG) {
          const float t = (one_group || row0 + dr < boundary) ? rg0 : rg1;
          v = mul ? v * t : v + t;
        }
        v += cb;
        if (relu) v = fmaxf(v, 0.f);
        *reinterpret_cast<uint16_t*>(scr + (32 * i + 4 * h + dr) * PITCH + (32 * j + l31) * 2) = epi_bf16(v);
      }
    }
  }
  asm volatile("" ::: "memory");     // the wave's LDS accesses execute in order; keep the compiler's order as well
  const __amdgpu_buffer_rsrc_t ro = buf_rsrc(pe.Cb + (int64_t)(m0 + wm * Cfg::WM) * pe.ldc + n0 + wn * Cfg::WN);
#pragma unroll
  for (int q = 0; q < Cfg::WM * PITCH / 1024; ++q) {
    const int byte = q * 1024 + lane * 16;
    const int row = byte / PITCH, inrow = byte % PITCH;
    const float4 v = *reinterpret_cast<const float4*>(scr + byte);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, (int)((uint32_t)row * (uint32_t)pe.ldc * 2u + inrow), 0, 0);
  }
}

static __global__ void splitk_reduce_kernel(EpiParams pe, int splits) {
  const int64_t total = (int64_t)pe.M * pe.N;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += pe.slab[(int64_t)s * total + e];
    const int row = (int)(e / pe.N), col = (int)(e - (int64_t)row * pe.N);
    if (pe.aux) pe.aux[(int64_t)row * pe.ldc + col] = v;
    const float o = epi_apply(pe, v, row, col);
    if (pe.Cb) pe.Cb[(int64_t)row * pe.ldc + col] = epi_bf16(o); else pe.C[(int64_t)row * pe.ldc + col] = o;
  }
}

// N % 4 == 0: 16-byte slab reads; a block covers 64 float4 outputs with 4 thread groups that each take every 4th
// split, so all the loads of a thread are in flight at once (the scalar kernel above is a chain of dependent
// 4-byte loads: 16 us for an 8-split 256 x 1024 output), and the groups combine through LDS in a fixed order.
static __global__ __launch_bounds__(256) void splitk_reduce4_kernel(EpiParams pe, int splits) {
  __shared__ float4 part[4][64];
  const int64_t total4 = (int64_t)pe.M * pe.N / 4;
  const int n4 = pe.N / 4;
  const float4* slab = reinterpret_cast<const float4*>(pe.slab);
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  for (int64_t e0 = (int64_t)blockIdx.x * 64; e0 < total4; e0 += (int64_t)gridDim.x * 64) {
    const int64_t e = e0 + lane;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < total4) {
      int s = grp;
      for (; s + 4 < splits; s += 8) {          // two independent loads per trip
        const float4 a = slab[(int64_t)s * total4 + e], b = slab[(int64_t)(s + 4) * total4 + e];
        v.x = (v.x + a.x) + b.x; v.y = (v.y + a.y) + b.y; v.z = (v.z + a.z) + b.z; v.w = (v.w + a.w) + b.w;
      }
      if (s < splits) {
        const float4 a = slab[(int64_t)s * total4 + e];
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
    }
    part[grp][lane] = v;
    __syncthreads();
    if (grp == 0 && e < total4) {
      const float4 p1 = part[1][lane], p2 = part[2][lane], p3 = part[3][lane];
      const float r[4] = {(v.x + p1.x) + (p2.x + p3.x), (v.y + p1.y) + (p2.y + p3.y),
                          (v.z + p1.z) + (p2.z + p3.z), (v.w + p1.w) + (p2.w + p3.w)};
      const int row = (int)(e / n4), col = 4 * (int)(e - (int64_t)row * n4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (pe.aux) pe.aux[(int64_t)row * pe.ldc + col + k] = r[k];
        const float o = epi_apply(pe, r[k], row, col + k);
        if (pe.Cb) pe.Cb[(int64_t)row * pe.ldc + col + k] = epi_bf16(o); else pe.C[(int64_t)row * pe.ldc + col + k] = o;
      }
    }
    __syncthreads();
  }
}


// Tile / split-K plan of a plain GEMM (gemm.hip); bk = K-step depth in elements (32 fp32, 64 bf16).
struct GemmPlan { int big; int tiles_m, tiles_n, nk, splits, ks_per_split, order; };
GemmPlan plan_gemm(int M, int N, int K, int bk = BK);

// Launch the reduction of split-K slabs (pe.slab [splits][M][N]) with the epilogue applied.
static inline int launch_splitk_reduce(const EpiParams& pe, int splits, hipStream_t s) {
  const bool vec = pe.N % 4 == 0;
  const int64_t total = (int64_t)pe.M * pe.N / (vec ? 4 : 1);
  int blocks = (int)((total + (vec ? 63 : 255)) / (vec ? 64 : 256));
  if (blocks > 4096) blocks = 4096;
  if (vec) hipLaunchKernelGGL(splitk_reduce4_kernel, dim3(blocks), dim3(256), 0, s, pe, splits);
  else hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, pe, splits);
  return check_hip(hipGetLastError(), "splitk_reduce launch");
}

}  // namespace vqa
