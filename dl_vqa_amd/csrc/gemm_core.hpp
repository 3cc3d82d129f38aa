// fp32 MFMA GEMM engine for gfx950 (CDNA4), shared by every contraction on the VQA hot path.
//
//   C[M,N] = sum_k A(m,k) * B(k,n)       exact fp32 (v_mfma_f32_32x32x2_f32)
//
// Design (see DESIGN.md "GEMM engine"):
//   * A workgroup = 4 (or 8) MFMA waves + 4 loader waves.  Each MFMA wave owns a (WM x WN) block of 32x32
//     MFMA tiles; accumulators stay in registers for the whole K loop.  BK = 32 per barrier.
//   * Operands are staged through LDS, double buffered, in an image chosen per operand type (LdsImage):
//     16-byte stores for both types, 16-byte fragment reads for k-contiguous operands, stride-1 4-byte
//     reads for reduction-major ones; all conflict-free.  The K order inside a K-step is permuted
//     (k = 8t + 4h + q) so that one 16-byte read feeds four consecutive MFMAs.
//   * Two kinds of operand loader fill those images from global memory, both with 16-byte per-lane loads,
//     8 lanes covering one 128-byte line:
//       type R ("k-contiguous rows"):  thread -> (row r, k-chunk c)
//       type C ("reduction-major"):    thread -> (k-row kr, m-chunk c)
//     Row addresses come from a functor, which is how implicit-im2col (conv forward), the
//     max-pool-routed gradient (conv dgrad / wgrad) and plain matrices share one main loop.
//   * A loader has two phases: issue(ks) computes addresses and issues UNCONDITIONAL 16-byte loads
//     (out-of-range rows/chunks are clamped to a valid address) into a Raw register set; finish()
//     applies masks / arg-max routing when the data is written to LDS one or more K-steps later
//     (a predicated load makes hipcc branch and wait per load).
#pragma once
#include <type_traits>
#include "common.hpp"

namespace vqa {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef VQA_DIAG
// Diagnostic build only (python -m dl_vqa_amd.build --diag -> libvqa_hip_diag.so, never loaded by default):
// MFMA waves stamp s_memtime around the MFMA block and around the barrier of every K-step.
// [0] cycles in mma_steps, [1] cycles waiting at the K-step barrier, [2] wave count, [3] total cycles
static __device__ unsigned long long vqa_diag_buf[4];
// loader waves: [0] finish+ds_write (incl. waiting for the loads), [1] issue, [2] barrier wait, [3] waves
static __device__ unsigned long long vqa_diag_ld[4];
#endif

// wave priorities of the two roles (s_setprio), overridable for experiments
#ifndef VQA_PRIO_MFMA
#define VQA_PRIO_MFMA 3
#endif
#ifndef VQA_EARLY_BARRIER
#define VQA_EARLY_BARRIER 2   // K-step barrier in front of the last fragment group (2: last two for 64x64 tiles)
#endif
#ifndef VQA_PF128
#define VQA_PF128 2   // K-steps of loads in flight for tiles up to 128x128
#endif
#ifndef VQA_PRIO_LOADER
#define VQA_PRIO_LOADER 0
#endif

constexpr int BK = 32;  // K-step depth (floats): 8 lanes x 16 B = one 128-B line per row

template <int BM_, int BN_, int WAVES_M_, int WAVES_N_, int LOADER_WAVES_ = 4, int PREFETCH_ = 0>
struct TileCfg {
  static_assert(LOADER_WAVES_ == 4 || LOADER_WAVES_ == 8, "4 or 8 loader waves");
  static constexpr int LT = 64 * LOADER_WAVES_;            // loader threads
  static_assert(WAVES_M_ * WAVES_N_ == 4 || WAVES_M_ * WAVES_N_ == 8, "4 or 8 MFMA waves per workgroup");
  static constexpr int BM = BM_, BN = BN_;
  static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_;
  static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  static constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1 && WM % 32 == 0 && WN % 32 == 0, "wave tile = 32x32 MFMA tiles");
  static constexpr int NVA = BM * 8 / LT, NVB = BN * 8 / LT;  // float4 per loader thread per K-step
  // One LDS buffer per operand per stage.  Its image depends on the operand's loader type (LdsImage):
  // both forms fit in 36 floats per row/column of the tile.
  static constexpr int ABUF_MAX = 36 * BM, BBUF_MAX = 36 * BN;      // floats, upper bound (SmemLayout has the exact sizes)
  static constexpr int SMEM_FLOATS = 2 * (ABUF_MAX + BBUF_MAX);
  static constexpr int SMEM_BYTES = SMEM_FLOATS * 4;
  // waves per SIMD the register allocator must leave room for (2nd __launch_bounds__ argument):
  // two 512-thread workgroups per CU (4 waves/SIMD, <= 128 VGPRs) up to 128x128, one above that
  static constexpr int NMFMA = WAVES_M * WAVES_N;          // MFMA waves (threads 0 .. 64*NMFMA-1)
  static constexpr int MFMA_THREADS = 64 * NMFMA;
  static constexpr int THREADS = MFMA_THREADS + LT;        // + the loader waves
  static constexpr int MIN_WAVES = (THREADS / 256) * ((SMEM_BYTES > 80 * 1024) ? 1 : 2);
  static_assert(NVA >= 1 && NVB >= 1, "tile too small for this many loader threads");
  // K-steps of global loads a loader thread keeps in flight (register ring).  A 64x64 tile's K-step is
  // only 1024 MFMA cycles (~0.45 us), shorter than an L2 round trip, so it needs 3 steps of run-ahead;
  // the big tiles' K-steps (>= 4096 cycles) cover the latency with one and have no registers to spare.
  static constexpr int PREFETCH = PREFETCH_ ? PREFETCH_ : (BM * BN <= 64 * 64) ? 3 : (BM * BN <= 128 * 128) ? VQA_PF128 : 1;
  // the persistent-tile loop re-initialises its loaders at tile seams and needs a single Raw set
  using Persistent = TileCfg<BM_, BN_, WAVES_M_, WAVES_N_, LOADER_WAVES_, 1>;
};

// LDS images.  An fp32 MFMA fragment is one dword per lane and the K order inside a K-step is free as
// long as A and B agree, so MFMA q (0..3) of group t (0..3) takes k = 8t + 4h + q (h = lane >> 5):
//   type R operand (k-contiguous rows in memory): image [row][RS = 36]; a thread's 16-byte chunk is ONE
//     ds_write_b128 (8 lanes = one 128-B row segment), and a lane fetches the four k of a group with ONE
//     ds_read_b128 at row*36 + 8t + 4h: 16 rows x 36-dword stride cover the 64 banks exactly once.
//   type C operand (reduction-major in memory): image [k][CS = tile + 4]; ONE ds_write_b128 per chunk
//     and stride-1 ds_read_b32 over the 32 lanes at row k = 8t + 4h + q.
// Both are conflict-free, and both need 4x fewer LDS write instructions than a transposing b32 image.
constexpr int LDS_RS = 36;
template <int TILE> struct LdsImage { static constexpr int CS = TILE + 4; };
// Exact LDS footprint of a tile for the operand types in use (floats per stage / bytes in total); a type C
// image is smaller than 36 * TILE.  (A = R, B = C 128x64 tiles take 54272 bytes and 79 VGPRs, enough for
// three workgroups per CU; with the launch bounds asking for 6 waves per SIMD conv1 dgrad ran 16 % SLOWER,
// 4.67 against 4.03 ms on the same box, so the kernels keep asking for two.)
template <class Cfg, bool AR, bool BR>
struct SmemLayout {
  static constexpr int ABUF = AR ? LDS_RS * Cfg::BM : BK * LdsImage<Cfg::BM>::CS;
  static constexpr int BBUF = BR ? LDS_RS * Cfg::BN : BK * LdsImage<Cfg::BN>::CS;
  static constexpr int BYTES = 2 * (ABUF + BBUF) * 4;
  static constexpr int WG_PER_CU = BYTES > 80 * 1024 ? 1 : 2;   // resident workgroups
};

// Loader thread -> staging coordinates, for LT = 256 or 512 loader threads (8 lanes = one 128-B line).
//   type R: rows (ltid >> 3) + (LT/8)*p, k-chunk ltid & 7 (k = 4*chunk .. +3)
//   type C: k-row (ltid >> 3) & 31, tile chunks (ltid & 7) + 8*(ltid >> 8) + (LT/32)*p  (elements 4*chunk .. +3)
template <int LT>
struct StageMap {
  static constexpr int RP = LT / 8;      // type R rows per pass
  static constexpr int CP = LT / 32;     // type C chunks per pass
  static __device__ __forceinline__ int r_row(int ltid, int p) { return (ltid >> 3) + RP * p; }
  static __device__ __forceinline__ int r_chunk(int ltid) { return ltid & 7; }
  static __device__ __forceinline__ int c_krow(int ltid) { return (ltid >> 3) & 31; }
  static __device__ __forceinline__ int c_chunk(int ltid, int p) { return (ltid & 7) + 8 * (ltid >> 8) + CP * p; }
};
template <int TILE, int NV, int LT>
__device__ __forceinline__ void lds_store_R(float* s, const float4 (&r)[NV], int ltid) {
#pragma unroll
  for (int p = 0; p < NV; ++p)
    *reinterpret_cast<float4*>(s + StageMap<LT>::r_row(ltid, p) * LDS_RS + 4 * StageMap<LT>::r_chunk(ltid)) = r[p];
}
template <int TILE, int NV, int LT>
__device__ __forceinline__ void lds_store_C(float* s, const float4 (&r)[NV], int ltid) {
#pragma unroll
  for (int p = 0; p < NV; ++p)
    *reinterpret_cast<float4*>(s + StageMap<LT>::c_krow(ltid) * LdsImage<TILE>::CS + 4 * StageMap<LT>::c_chunk(ltid, p)) = r[p];
}

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// Zero the elements of a 16-byte chunk that lie at or beyond `len` (chunk starts at element e0).
__device__ __forceinline__ float4 mask4(float4 v, bool on, int e0, int len) {
  v.x = (on && e0 < len) ? v.x : 0.f;
  v.y = (on && e0 + 1 < len) ? v.y : 0.f;
  v.z = (on && e0 + 2 < len) ? v.z : 0.f;
  v.w = (on && e0 + 3 < len) ? v.w : 0.f;
  return v;
}

// ---------------------------------------------------------------- buffer addressing
// Loader waves share their SIMD with MFMA waves, and on gfx950 every VALU instruction a loader wave issues
// takes the slot of an MFMA pass (measured: kernel time ~= MFMA-only time + loader VALU time, whatever the
// wave priorities), while SALU, VMEM and LDS instructions issue beside the MFMA stream for free.  So the
// loaders keep their per-K-step work off the VALU: a thread's offsets inside the tile are computed once
// (init), the K-step advance is a scalar add on the buffer resource's base address, and rows / columns
// past the end are lanes whose offset is BUF_OOB: the hardware range check returns zeros for them, which
// replaces both the address clamp and the v_cndmask masking of a plain global load.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// BUF_OOB (+ any immediate offset, no 32-bit wrap) >= num_records: the lane reads zeros, no memory access
constexpr uint32_t BUF_OOB = 0xffff0000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* base, uint32_t bytes = 0xffff0000u) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff = 0) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ uint32_t buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff = 0) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0);
}

__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, float v, uint32_t voff, uint32_t soff = 0) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, uint8_t v, uint32_t voff, uint32_t soff = 0) {
  __builtin_amdgcn_raw_buffer_store_b8(v, r, (int)voff, (int)soff, 0);
}

// ---------------------------------------------------------------- plain matrix loaders
// Both require ld >= roundup4(row length) (a 16-byte load never leaves the row's allocation) and
// 256 * ld * 4 < 2^32 (offsets inside a tile are 32-bit; the tile origin is a 64-bit scalar).
// Loader concept:  init(params, tile origin, loader thread, first K-step)
//                  issue(ks, Raw&)   global loads of K-step ks into registers (any ks >= first; past the
//                                    end of K every lane is out of range and reads zeros)
//                  finish(Raw, out)  the NV 16-byte chunks as they go to LDS
// Type R over a row-major matrix X[rows][K]: used for A=[M][K] and B=[N][K].
template <int NV, int LT = 256>
struct PlainR {
  struct Params { const float* p; int64_t ld; int rows; int K; };
  struct Raw { float4 v[NV]; int rem; };   // rem = K - ks*BK (uniform): < BK on the tail step, <= 0 past the end
  static constexpr bool kTypeR = true;
  const float* base;      // first row of the tile (uniform)
  uint32_t voff[NV];      // byte offset of the thread's chunk from base; BUF_OOB for rows past the end
  int K, c4;
  __device__ __forceinline__ void init(const Params& q, int row0, int tid, int /*ks0*/) {
    K = q.K; c4 = 4 * StageMap<LT>::r_chunk(tid);
    base = q.p + (int64_t)row0 * q.ld;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int rl = StageMap<LT>::r_row(tid, p);
      voff[p] = row0 + rl < q.rows ? (uint32_t)(rl * (int)q.ld + c4) * 4u : BUF_OOB;
    }
  }
  __device__ __forceinline__ void issue(int ks, Raw& r) {
    r.rem = K - ks * BK;
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(base + (int64_t)ks * BK);
    if (r.rem >= BK) {
#pragma unroll
      for (int p = 0; p < NV; ++p) r.v[p] = buf_load16(rs, voff[p]);
    } else {          // tail K-step (or past the end): chunks at or beyond K read zeros
#pragma unroll
      for (int p = 0; p < NV; ++p) r.v[p] = buf_load16(rs, c4 < r.rem ? voff[p] : BUF_OOB);
    }
  }
  __device__ __forceinline__ void finish(const Raw& r, float4 (&o)[NV]) const {
    if (r.rem >= BK) {
#pragma unroll
      for (int p = 0; p < NV; ++p) o[p] = r.v[p];
    } else {          // a chunk straddling K (K % 4 != 0) keeps only its elements below K
#pragma unroll
      for (int p = 0; p < NV; ++p) o[p] = mask4(r.v[p], true, c4, r.rem);
    }
  }
};

// Type C over a row-major matrix X[K][cols] (reduction index is the slow one).  Columns at or beyond
// `cols` inside a straddling chunk are not masked: they only reach accumulator rows / columns that no
// epilogue stores.
template <int NV, int LT = 256>
struct PlainC {
  struct Params { const float* p; int64_t ld; int cols; int K; };
  struct Raw { float4 v[NV]; };
  static constexpr bool kTypeR = false;
  const float* base;      // column col0 of row 0 (uniform)
  int64_t ld;
  uint32_t voff[NV];
  int K, kr;
  __device__ __forceinline__ void init(const Params& q, int col0, int tid, int /*ks0*/) {
    ld = q.ld; K = q.K; kr = StageMap<LT>::c_krow(tid);
    base = q.p + col0;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int cl = 4 * StageMap<LT>::c_chunk(tid, p);
      voff[p] = col0 + cl < q.cols ? (uint32_t)(kr * (int)q.ld + cl) * 4u : BUF_OOB;
    }
  }
  __device__ __forceinline__ void issue(int ks, Raw& r) {
    const int rem = K - ks * BK;
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(base + (int64_t)ks * BK * ld);
    if (rem >= BK) {
#pragma unroll
      for (int p = 0; p < NV; ++p) r.v[p] = buf_load16(rs, voff[p]);
    } else {
#pragma unroll
      for (int p = 0; p < NV; ++p) r.v[p] = buf_load16(rs, kr < rem ? voff[p] : BUF_OOB);
    }
  }
  __device__ __forceinline__ void finish(const Raw& r, float4 (&o)[NV]) const {
#pragma unroll
    for (int p = 0; p < NV; ++p) o[p] = r.v[p];
  }
};

// ---------------------------------------------------------------- the main loop
// Wave-specialised: a workgroup is 512 threads = 4 MFMA waves + 4 loader waves (one of each per SIMD).
//   MFMA waves   : per K-step, 16 k2-steps of ds_read_b32 fragment reads + TM*TN MFMAs each; nothing else.
//   loader waves : per K-step, finish() + ds_write the tile of step k+1 (loaded during step k-1) into the
//                  other LDS buffer, then issue() the global loads of step k+2 into registers.
//   one __syncthreads() per K-step joins the two roles.
// Why: on gfx950 a wave issues in order, so every global_load / ds_write / address instruction a wave
// executes is a hole in ITS MFMA stream (tools/mfma_peak.hip: 8 global_load_dwordx4 per K-step cost a
// 4-wave workgroup 14-20 % of the MFMA rate even when they hit L2).  With the staging work on other
// waves of the same SIMD, the MFMA waves sustain the bare ds_read + MFMA + barrier loop (96-98 % of
// peak in the same micro-benchmark).  A global load has a full K-step (>= 4096 MFMA cycles) to land.
// The two roles are separate loops (not one loop with a role test) so that each gets its own register
// allocation: accumulators + fragments for one, Raw tiles + addresses for the other, both <= 128 VGPRs.

// NG groups of 4 k2-steps from one stage; the fragments of group t+1 are fetched before the MFMAs of group t.
// CS: also accumulate the column sums of the B operand into cs[TN] (one value per lane = its column, over the
// k the lane's fragments hold): the conv wgrad bias gradient, taken from fragments the wave reads anyway.
template <class Cfg, bool AR, bool BR, int NG, bool CS = false>
__device__ __forceinline__ void mma_steps(const float* As, const float* Bs, f32x16 (&acc)[Cfg::TM][Cfg::TN],
                                          int wm, int wn, int lane, float* cs = nullptr) {
  const int l31 = lane & 31, h = lane >> 5;
  constexpr int CSA = LdsImage<Cfg::BM>::CS, CSB = LdsImage<Cfg::BN>::CS;
  // group t, MFMA q: k = 8t + 4h + q
  const float* ap = AR ? As + (wm * Cfg::WM + l31) * LDS_RS + 4 * h : As + 4 * h * CSA + wm * Cfg::WM + l31;
  const float* bp = BR ? Bs + (wn * Cfg::WN + l31) * LDS_RS + 4 * h : Bs + 4 * h * CSB + wn * Cfg::WN + l31;
  float a[2][4][Cfg::TM], b[2][4][Cfg::TN];
  auto fetch = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) {
      if (AR) {
        const float4 v = *reinterpret_cast<const float4*>(ap + 32 * i * LDS_RS + 8 * t);
        a[buf][0][i] = v.x; a[buf][1][i] = v.y; a[buf][2][i] = v.z; a[buf][3][i] = v.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[buf][q][i] = ap[(8 * t + q) * CSA + 32 * i];
      }
    }
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      if (BR) {
        const float4 v = *reinterpret_cast<const float4*>(bp + 32 * j * LDS_RS + 8 * t);
        b[buf][0][j] = v.x; b[buf][1][j] = v.y; b[buf][2][j] = v.z; b[buf][3][j] = v.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) b[buf][q][j] = bp[(8 * t + q) * CSB + 32 * j];
      }
    }
  };
  fetch(0, 0);
#pragma unroll
  for (int t = 0; t < NG; ++t) {
    const int cur = t & 1;
    if (t + 1 < NG) fetch(t + 1, cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);   // reads of group t+1 stay above the MFMAs of group t
    if (CS) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) cs[j] += b[cur][q][j];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][q][i], b[cur][q][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <class Cfg, class L, bool IS_A>
__device__ __forceinline__ void stage_store_one(const L& ld, const typename L::Raw& raw, float* dst, int ltid) {
  constexpr int NV = IS_A ? Cfg::NVA : Cfg::NVB;
  constexpr int TILE = IS_A ? Cfg::BM : Cfg::BN;
  float4 r[NV];
  ld.finish(raw, r);
#ifdef VQA_EXP_SKIP_STORE   // timing experiment: keep the global loads and finish(), drop the LDS writes
#pragma unroll
  for (int p = 0; p < NV; ++p) asm volatile("" ::"v"(r[p].x), "v"(r[p].y), "v"(r[p].z), "v"(r[p].w));
  (void)dst; (void)ltid;
#else
  if (L::kTypeR) lds_store_R<TILE, NV, Cfg::LT>(dst, r, ltid); else lds_store_C<TILE, NV, Cfg::LT>(dst, r, ltid);
#endif
}

template <class Cfg> __device__ __forceinline__ bool is_loader_wave() { return threadIdx.x >= Cfg::MFMA_THREADS; }
// staging thread id (0..255): the loader waves are the last four of the workgroup; MFMA waves get a
// harmless in-range value (they construct loaders but never use them)
template <class Cfg> __device__ __forceinline__ int loader_tid() { return (threadIdx.x - Cfg::MFMA_THREADS) & (Cfg::LT - 1); }

// Loader role: K-steps [ks0, ks1).  Loaders tolerate issue() past the end (addresses clamped, data masked).
// Ring of D = Cfg::PREFETCH Raw register sets: at K-step ks the tile of step ks+1 is written to the other
// LDS buffer and the loads of step ks+1+D are issued into the set it frees.
template <class Cfg, class AL, class BL>
__device__ __forceinline__ void loader_loop(AL& al, BL& bl, int ks0, int ks1, float* smem) {
  constexpr int D = Cfg::PREFETCH;
  using SL = SmemLayout<Cfg, AL::kTypeR, BL::kTypeR>;
  const int ltid = loader_tid<Cfg>();
  float* const As0 = smem;
  float* const Bs0 = smem + 2 * SL::ABUF;
  typename AL::Raw rawA[D];
  typename BL::Raw rawB[D];
  if (VQA_PRIO_LOADER) __builtin_amdgcn_s_setprio(VQA_PRIO_LOADER);
  al.issue(ks0, rawA[0]);
  bl.issue(ks0, rawB[0]);
  stage_store_one<Cfg, AL, true>(al, rawA[0], As0, ltid);
  stage_store_one<Cfg, BL, false>(bl, rawB[0], Bs0, ltid);
#pragma unroll
  for (int d = 0; d < D; ++d) {
    al.issue(ks0 + 1 + d, rawA[d]);
    bl.issue(ks0 + 1 + d, rawB[d]);
  }
  __syncthreads();
#ifdef VQA_DIAG
  unsigned long long t_st = 0, t_is = 0, t_ba = 0;
#endif
  for (int ks = ks0; ks < ks1; ks += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (ks + d < ks1) {                                 // block-uniform; one barrier per K-step, as the MFMA role
        const int nxt = ((ks + d - ks0) & 1) ^ 1;
#ifdef VQA_DIAG
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        stage_store_one<Cfg, AL, true>(al, rawA[d], As0 + nxt * SL::ABUF, ltid);
        stage_store_one<Cfg, BL, false>(bl, rawB[d], Bs0 + nxt * SL::BBUF, ltid);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        al.issue(ks + d + 1 + D, rawA[d]);
        bl.issue(ks + d + 1 + D, rawB[d]);
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
        t_st += t1 - t0; t_is += t2 - t1; t_ba += t3 - t2;
#elif defined(VQA_EXP_SKIP_LOAD)   // timing experiments only: the MFMA side alone
        __syncthreads();
#elif defined(VQA_EXP_STORE_ONLY)  // timing experiments only: LDS stores of stale registers, no global loads
        stage_store_one<Cfg, AL, true>(al, rawA[d], As0 + nxt * SL::ABUF, ltid);
        stage_store_one<Cfg, BL, false>(bl, rawB[d], Bs0 + nxt * SL::BBUF, ltid);
        __syncthreads();
#else
        stage_store_one<Cfg, AL, true>(al, rawA[d], As0 + nxt * SL::ABUF, ltid);
        al.issue(ks + d + 1 + D, rawA[d]);
        stage_store_one<Cfg, BL, false>(bl, rawB[d], Bs0 + nxt * SL::BBUF, ltid);
        bl.issue(ks + d + 1 + D, rawB[d]);
        __syncthreads();
#endif
      }
    }
  }
#ifdef VQA_DIAG
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&vqa_diag_ld[0], t_st); atomicAdd(&vqa_diag_ld[1], t_is); atomicAdd(&vqa_diag_ld[2], t_ba);
    atomicAdd(&vqa_diag_ld[3], 1ULL);
  }
#endif
}

// MFMA role.  SHORT_TAIL (conv0 forward, K = 36 = 32 + 4): the last K-step runs 4 k2-steps instead of 16.
template <class Cfg, bool AR, bool BR, bool SHORT_TAIL, bool CS = false>
__device__ __forceinline__ void mfma_loop(f32x16 (&acc)[Cfg::TM][Cfg::TN], int ks0, int ks1, int Ktot,
                                          const float* smem, float* cs = nullptr) {
  using SL = SmemLayout<Cfg, AR, BR>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const float* const As0 = smem;
  const float* const Bs0 = smem + 2 * SL::ABUF;
  __builtin_amdgcn_s_setprio(VQA_PRIO_MFMA);   // MFMA waves win issue arbitration over the loader waves of their SIMD
#ifdef VQA_DIAG
  unsigned long long t_mma = 0, t_bar = 0;
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();
  for (int ks = ks0; ks < ks1; ++ks) {
    const int cur = (ks - ks0) & 1;
    const float* const Ac = As0 + cur * SL::ABUF;
    const float* const Bc = Bs0 + cur * SL::BBUF;
#ifdef VQA_DIAG
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
#ifndef VQA_EXP_SKIP_MFMA   // timing experiments only (tools/build_variant.sh): results are garbage
    if (SHORT_TAIL && Ktot - ks * BK <= 8) mma_steps<Cfg, AR, BR, 1, CS>(Ac, Bc, acc, wm, wn, lane, cs);
    else mma_steps<Cfg, AR, BR, 4, CS>(Ac, Bc, acc, wm, wn, lane, cs);
#endif
#ifdef VQA_DIAG
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef VQA_DIAG
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    t_mma += t1 - t0; t_bar += t2 - t1;
#endif
  }
#ifdef VQA_DIAG
  if (lane == 0) {
    atomicAdd(&vqa_diag_buf[0], t_mma); atomicAdd(&vqa_diag_buf[1], t_bar); atomicAdd(&vqa_diag_buf[2], 1ULL);
    atomicAdd(&vqa_diag_buf[3], __builtin_amdgcn_s_memtime() - t_begin);
  }
#endif
}

// MFMA role with the K-step barrier moved in front of the LAST fragment group.  By then every fragment of the
// stage is in registers (group 3 was fetched under group 2's MFMAs), so the stage can be handed back to the
// loaders a quarter of a K-step early, and the first fragments of the NEXT stage are fetched under group 3's
// MFMAs instead of after the barrier with the MFMA pipe idle.  Same number of barriers as mfma_loop.
template <class Cfg, bool AR, bool BR, bool CS>
__device__ __forceinline__ void mfma_loop_eb(f32x16 (&acc)[Cfg::TM][Cfg::TN], int ks0, int ks1, const float* smem,
                                             float* cs) {
  using SL = SmemLayout<Cfg, AR, BR>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const int l31 = lane & 31, h = lane >> 5;
  constexpr int CSA = LdsImage<Cfg::BM>::CS, CSB = LdsImage<Cfg::BN>::CS;
  const float* const As0 = smem + (AR ? (wm * Cfg::WM + l31) * LDS_RS + 4 * h : 4 * h * CSA + wm * Cfg::WM + l31);
  const float* const Bs0 = smem + 2 * SL::ABUF + (BR ? (wn * Cfg::WN + l31) * LDS_RS + 4 * h : 4 * h * CSB + wn * Cfg::WN + l31);
  // 64x64 tiles (one MFMA tile per wave, 16 MFMAs per K-step) put the barrier in front of the last TWO groups:
  // four fragment buffers of 8 registers, the stage goes back to the loaders half a K-step early
  constexpr bool EB2 = Cfg::TM * Cfg::TN == 1 && VQA_EARLY_BARRIER >= 2;   // (no gain on the 96x128 wgrad tile)
  float a[EB2 ? 4 : 2][4][Cfg::TM], b[EB2 ? 4 : 2][4][Cfg::TN];
  auto fetch = [&](const float* ap, const float* bp, int t, int buf) {
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) {
      if (AR) {
        const float4 v = *reinterpret_cast<const float4*>(ap + 32 * i * LDS_RS + 8 * t);
        a[buf][0][i] = v.x; a[buf][1][i] = v.y; a[buf][2][i] = v.z; a[buf][3][i] = v.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[buf][q][i] = ap[(8 * t + q) * CSA + 32 * i];
      }
    }
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      if (BR) {
        const float4 v = *reinterpret_cast<const float4*>(bp + 32 * j * LDS_RS + 8 * t);
        b[buf][0][j] = v.x; b[buf][1][j] = v.y; b[buf][2][j] = v.z; b[buf][3][j] = v.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) b[buf][q][j] = bp[(8 * t + q) * CSB + 32 * j];
      }
    }
  };
  auto mma = [&](int buf) {
    if (CS) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) cs[j] += b[buf][q][j];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][q][i], b[buf][q][j], acc[i][j], 0, 0, 0);
  };
  __builtin_amdgcn_s_setprio(VQA_PRIO_MFMA);
  __syncthreads();
  if (EB2) {
    // buffers 0,1 hold groups 0,1 of the current stage on entry to a K-step; 2,3 take groups 2,3
    fetch(As0, Bs0, 0, 0);
    fetch(As0, Bs0, 1, 1);
    for (int ks = ks0; ks < ks1; ++ks) {
      const int cur = (ks - ks0) & 1;
      const float* const Ac = As0 + cur * SL::ABUF;
      const float* const Bc = Bs0 + cur * SL::BBUF;
      const float* const An = As0 + (cur ^ 1) * SL::ABUF;
      const float* const Bn = Bs0 + (cur ^ 1) * SL::BBUF;
      fetch(Ac, Bc, 2, 2);
      __builtin_amdgcn_sched_barrier(0);
      mma(0);
      __builtin_amdgcn_sched_barrier(0);
      fetch(Ac, Bc, 3, 3);
      __builtin_amdgcn_sched_barrier(0);
      mma(1);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();   // groups 2, 3 are in registers: the stage is free; the next stage is complete
      if (ks + 1 < ks1) fetch(An, Bn, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma(2);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < ks1) fetch(An, Bn, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(3);
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  fetch(As0, Bs0, 0, 0);
  for (int ks = ks0; ks < ks1; ++ks) {
    const int cur = (ks - ks0) & 1;
    const float* const Ac = As0 + cur * SL::ABUF;
    const float* const Bc = Bs0 + cur * SL::BBUF;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      fetch(Ac, Bc, t + 1, (t + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);   // reads of group t+1 stay above the MFMAs of group t
      mma(t & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // every fragment of this stage is in registers; the next stage is complete
    if (ks + 1 < ks1) fetch(As0 + (cur ^ 1) * SL::ABUF, Bs0 + (cur ^ 1) * SL::BBUF, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma(1);            // group 3, under the next stage's first fragment reads
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Whole contraction over K-steps [ks0, ks1) (ks1 > ks0).  Returns true for MFMA waves (which hold the
// accumulators and run the epilogue); loader waves return false and must do nothing further that
// needs a barrier.  The loaders live entirely inside the loader branch -- init(al, bl) sets them up,
// done(al, bl) runs after the last K-step -- so that MFMA waves spend no VALU on them and, as important,
// so that their wave-uniform state stays in SGPRs (a value defined under the role test and used after
// the join would count as divergent).
// cs != nullptr (uniform): the MFMA waves also accumulate B's column sums (mma_steps CS) into cs[TN].
template <class Cfg, class AL, class BL, bool SHORT_TAIL = false, class Init, class Done>
__device__ __forceinline__ bool gemm_mainloop(Init&& init, Done&& done, f32x16 (&acc)[Cfg::TM][Cfg::TN], int ks0,
                                              int ks1, int Ktot, float* smem, float* cs = nullptr) {
  if (is_loader_wave<Cfg>()) {
    AL al; BL bl;
    init(al, bl);
    loader_loop<Cfg>(al, bl, ks0, ks1, smem);
    done(al, bl);
    return false;
  }
#if VQA_EARLY_BARRIER && !defined(VQA_DIAG) && !defined(VQA_EXP_SKIP_MFMA)
  if (!SHORT_TAIL) {
    if (cs) mfma_loop_eb<Cfg, AL::kTypeR, BL::kTypeR, true>(acc, ks0, ks1, smem, cs);
    else mfma_loop_eb<Cfg, AL::kTypeR, BL::kTypeR, false>(acc, ks0, ks1, smem, nullptr);
    return true;
  }
#endif
  if (cs) mfma_loop<Cfg, AL::kTypeR, BL::kTypeR, SHORT_TAIL, true>(acc, ks0, ks1, Ktot, smem, cs);
  else mfma_loop<Cfg, AL::kTypeR, BL::kTypeR, SHORT_TAIL, false>(acc, ks0, ks1, Ktot, smem);
  return true;
}

// ---------------------------------------------------------------- persistent tiles
// A workgroup walks the tiles  first, first + stride, ...  (< ntiles), every tile running the SAME K-steps
// [0, nk).  The loader waves treat the tiles as one flat sequence of K-steps: while the MFMA waves run the
// epilogue of tile t, the first K-step of tile t+1 is already in LDS and its second in registers, and no
// workgroup is re-dispatched between tiles (a 128x128 tile with K = 256 lives only ~28 us, so dispatch +
// prologue + epilogue per tile were a large part of it).  The loader functors are re-initialised right
// before the first issue() of a new tile.
//   init_tile(tile, AL&, BL&)   : (re)initialise both loaders for a tile
//   epilogue(tile, acc)         : MFMA waves only, no barriers inside
// Barrier protocol (the invariant a workgroup's life depends on): BOTH roles execute exactly 1 + my_tiles * nk workgroup
// barriers -- the loader waves one after the prologue and one per stored stage, the MFMA waves one before the first stage
// and one per computed stage -- on paths that share no code.  A -DVQA_DIAG build counts them per role (bar_dbg) and
// tools/diag_barriers.py asserts the two averages are equal after a run.
template <class Cfg, class AL, class BL, bool SHORT_TAIL, class InitTile, class Epilogue>
__device__ __forceinline__ void gemm_persistent(int first, int stride, int ntiles, int nk, int Ktot, float* smem,
                                                InitTile&& init_tile, Epilogue&& epilogue,
                                                unsigned long long* bar_dbg = nullptr) {
  using SL = SmemLayout<Cfg, AL::kTypeR, BL::kTypeR>;
  constexpr int D = Cfg::PREFETCH;
  if (first >= ntiles) return;
  const int my_tiles = (ntiles - first + stride - 1) / stride;
  const int total = my_tiles * nk;
  if (is_loader_wave<Cfg>()) {
    const int ltid = loader_tid<Cfg>();
    float* const As0 = smem;
    float* const Bs0 = smem + 2 * SL::ABUF;
    AL al; BL bl;
    // ring of D register sets, as loader_loop; finish() of every loader reads only the Raw set and
    // thread constants, so re-initialising the loaders for the next tile while sets of the previous tile
    // are still waiting to be stored is safe
    typename AL::Raw rawA[D];
    typename BL::Raw rawB[D];
    int tile = first, ks = 0;            // the K-step the next issue() fetches
    init_tile(tile, al, bl);
    auto next = [&](typename AL::Raw& ra, typename BL::Raw& rb) {   // issue K-step (tile, ks), advance across tile seams
      if (ks == nk) {
        ks = 0;
        if (tile + stride < ntiles) tile += stride;     // past the last tile: harmless refetch, never read
        init_tile(tile, al, bl);
      }
      al.issue(ks, ra);
      bl.issue(ks, rb);
      ++ks;
    };
    next(rawA[0], rawB[0]);
    stage_store_one<Cfg, AL, true>(al, rawA[0], As0, ltid);
    stage_store_one<Cfg, BL, false>(bl, rawB[0], Bs0, ltid);
#pragma unroll
    for (int d = 0; d < D; ++d) next(rawA[d], rawB[d]);
    __syncthreads();
#ifdef VQA_DIAG
    unsigned long long nbar = 1;
#endif
    for (int s = 0; s < total; s += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (s + d < total) {
          const int nxt = ((s + d) & 1) ^ 1;
          stage_store_one<Cfg, AL, true>(al, rawA[d], As0 + nxt * SL::ABUF, ltid);
          stage_store_one<Cfg, BL, false>(bl, rawB[d], Bs0 + nxt * SL::BBUF, ltid);
          next(rawA[d], rawB[d]);
          __syncthreads();
#ifdef VQA_DIAG
          ++nbar;
#endif
        }
      }
    }
#ifdef VQA_DIAG
    if (bar_dbg && (threadIdx.x & 63) == 0) { atomicAdd(bar_dbg + 0, nbar); atomicAdd(bar_dbg + 1, 1ull); }
#endif
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const float* const As0 = smem;
  const float* const Bs0 = smem + 2 * SL::ABUF;
  __builtin_amdgcn_s_setprio(VQA_PRIO_MFMA);
  __syncthreads();
#ifdef VQA_DIAG
  unsigned long long nbar = 1;
#endif
  int s = 0;
  for (int tile = first; tile < ntiles; tile += stride) {
    f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int ks = 0; ks < nk; ++ks, ++s) {
      const float* const Ac = As0 + (s & 1) * SL::ABUF;
      const float* const Bc = Bs0 + (s & 1) * SL::BBUF;
      if (SHORT_TAIL && Ktot - ks * BK <= 8) mma_steps<Cfg, AL::kTypeR, BL::kTypeR, 1>(Ac, Bc, acc, wm, wn, lane);
      else mma_steps<Cfg, AL::kTypeR, BL::kTypeR, 4>(Ac, Bc, acc, wm, wn, lane);
      __syncthreads();
#ifdef VQA_DIAG
      ++nbar;
#endif
    }
    epilogue(tile, acc);
  }
#ifdef VQA_DIAG
  if (bar_dbg && lane == 0) { atomicAdd(bar_dbg + 2, nbar); atomicAdd(bar_dbg + 3, 1ull); }
#endif
}

template <class Cfg>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// Accumulator element (i, j, r) of this lane is C[row][col] with
//   row = wm*WM + 32*i + (r&3) + 8*(r>>2) + 4*(lane>>5),  col = wn*WN + 32*j + (lane&31)
// (C/D map of v_mfma_f32_32x32x2_f32; rows come from the A operand, columns from B).
template <class Cfg>
__device__ __forceinline__ int acc_row(int wm, int i, int r, int lane) {
  return wm * Cfg::WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
template <class Cfg>
__device__ __forceinline__ int acc_col(int wn, int j, int lane) {
  return wn * Cfg::WN + 32 * j + (lane & 31);
}

// Plain store of the accumulators to a row-major matrix out[rows][ld] (dgrad's dx, the split-K slabs).
// Buffer resource based at each 32x32 tile's first element: the lane's part (row 4*(lane>>5), column
// lane&31) is one VGPR for the whole epilogue, element r's row a scalar offset; interior tiles carry no
// predicate, on edge tiles an invalid element's offset is BUF_OOB and the hardware drops the store.
template <class Cfg>
__device__ __forceinline__ void store_acc_tiles(f32x16 (&acc)[Cfg::TM][Cfg::TN], float* out, int64_t ld, int rows,
                                                int cols, int m0, int n0, int wm, int wn, int lane) {
  const bool interior = m0 + Cfg::BM <= rows && n0 + Cfg::BN <= cols;       // uniform
  const uint32_t ldb = (uint32_t)ld * 4u;
  const uint32_t vl = (uint32_t)(4 * (lane >> 5)) * ldb + 4u * (uint32_t)(lane & 31);
  auto body = [&](auto inner) {
    constexpr bool INNER = decltype(inner)::value;
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) {
        const int rowt = m0 + wm * Cfg::WM + 32 * i, colt = n0 + wn * Cfg::WN + 32 * j;
        const __amdgpu_buffer_rsrc_t rs = buf_rsrc(out + (int64_t)rowt * ld + colt);
        const bool cok = colt + (lane & 31) < cols;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          const bool ok = INNER || (cok && rowt + 4 * (lane >> 5) + dr < rows);
          buf_store4(rs, acc[i][j][r], ok ? vl : BUF_OOB, (uint32_t)dr * ldb);
        }
      }
  };
  if (interior) body(std::true_type{}); else body(std::false_type{});
}

// Workgroup id -> (mt, nt, split).  The grid is one-dimensional; logical ids run nt fastest, then
// mt, then split, and xcd_swizzle gives each XCD a contiguous range of logical ids: workgroups that
// share an A panel (same mt) or the same reduction slice (same split) meet in one XCD's L2.
// order 1 ("weight stationary", for skinny GEMMs whose B operand is the big one, e.g. the LSTM's
// h[256 x 1024] . W_hh^T[1024 x 4096]): nt runs SLOWEST, so an XCD's contiguous id range covers only
// tiles_n/8 column tiles and re-reads just that 1/8 slice of the weights, which then fits its 4 MiB L2.
struct TileCoord { int mt, nt, split; };
__device__ __forceinline__ TileCoord tile_coord(int tiles_m, int tiles_n, int order = 0, int splits = 1) {
  const int t = xcd_swizzle(blockIdx.x, gridDim.x);
  const int tiles = tiles_m * tiles_n;
  TileCoord c;
  if (order == 1) {
    const int per_nt = tiles_m * splits;
    c.nt = t / per_nt;
    const int u = t - c.nt * per_nt;
    c.split = u / tiles_m;
    c.mt = u - c.split * tiles_m;
    return c;
  }
  c.split = t / tiles;
  const int u = t - c.split * tiles;
  c.mt = u / tiles_n;
  c.nt = u - c.mt * tiles_n;
  return c;
}

}  // namespace vqa
