// bf16 instantiation of the GEMM engine (gemm_core.hpp) for gfx950: v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//
// BASELINE configs[3] (bf16 conv/FC path).  Same wave-specialised structure as the fp32 engine -- loader waves stage
// operand tiles through double-buffered LDS with buffer loads, MFMA waves only read fragments and issue MFMAs -- and
// the same byte geometry: a K-step is 128 bytes of every k-contiguous row, i.e. 64 bf16 instead of 32 floats.
//   type R operand (k-contiguous rows):  the LDS image, the loaders and the fragment reads are the fp32 engine's,
//     byte for byte (a "float" is a pair of bf16): image [row][36 dwords]; the ds_read_b128 that fed four
//     32x32x2 fp32 MFMAs now is ONE 32x32x16 operand (lane (r, h): k = 16 m + 8 h + 0..7 of MFMA m = 0..3).
//   type C operand (reduction-major rows, e.g. both operands of a weight gradient): image [k 0..63][tile] of bf16 with
//     a row stride = 64 or 192 (mod 256) bytes, written with one ds_write_b128 per 8-element chunk and read with
//     ds_read_b64_tr_b16, the hardware transpose read: per 16-lane group a 4 (k) x 16 (tile) block comes back
//     column-major, so two reads give a lane its 8 consecutive k of one tile column.  Conflict-free by the stride.
// Accumulator layout, epilogues, XCD swizzle and split-K slabs are shared with the fp32 engine.
#pragma once
#include "gemm_core.hpp"

namespace vqa {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int BKB = 64;   // bf16 K-step depth in elements (= BK dwords of a k-contiguous row)

// fp32 -> bf16, round to nearest even (v_cvt_pk_bf16_f32; NaN stays NaN)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint16_t bf16_bits(float x) { return (uint16_t)(pack_bf16x2(x, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ void buf_store2(__amdgpu_buffer_rsrc_t r, uint16_t v, uint32_t voff, uint32_t soff = 0) {
  __builtin_amdgcn_raw_buffer_store_b16(v, r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ uint2 buf_load8(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff = 0) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
  return make_uint2(v.x, v.y);
}

// ---------------------------------------------------------------- type C image of bf16
template <int TILE>
struct LdsImageB {
  static constexpr int RAW = TILE * 2;                                                     // bytes of a k row
  static constexpr int RSB = RAW + ((RAW % 256 == 64 || RAW % 256 == 192) ? 0 : 64);      // row stride: 64 / 192 mod 256
  static constexpr int DWORDS = BKB * RSB / 4;                                             // one stage
  static_assert(TILE % 32 == 0, "tile sizes are multiples of 32");
};
template <class Cfg, bool AR, bool BR>
struct SmemLayoutB {
  static constexpr int ABUF = AR ? LDS_RS * Cfg::BM : LdsImageB<Cfg::BM>::DWORDS;   // dwords per stage
  static constexpr int BBUF = BR ? LDS_RS * Cfg::BN : LdsImageB<Cfg::BN>::DWORDS;
  static constexpr int BYTES = 2 * (ABUF + BBUF) * 4;
  static constexpr int WG_PER_CU = BYTES > 80 * 1024 ? 1 : 2;
};

// Loader thread -> staging coordinates of a type C bf16 tile: 64 k rows, LT/64 threads per row, 16-byte chunks
// (8 elements) chunk(p) = (ltid % TPR) + TPR * p.  NV = TILE / (8 * TPR) chunks per thread.
template <int TILE, int LT>
struct StageMapCb {
  static constexpr int TPR = LT / BKB;              // threads per k row (4 with 256 loader threads)
  static constexpr int NV = TILE / (8 * TPR);
  static_assert(TILE % (8 * TPR) == 0, "tile not divisible into chunks");
  static __device__ __forceinline__ int krow(int ltid) { return ltid / TPR; }
  static __device__ __forceinline__ int chunk(int ltid, int p) { return (ltid % TPR) + TPR * p; }
};
template <int TILE, int NV, int LT>
__device__ __forceinline__ void lds_store_Cb(float* s, const float4 (&r)[NV], int ltid) {
  char* base = reinterpret_cast<char*>(s) + StageMapCb<TILE, LT>::krow(ltid) * LdsImageB<TILE>::RSB;
#pragma unroll
  for (int p = 0; p < NV; ++p)
    *reinterpret_cast<float4*>(base + 16 * StageMapCb<TILE, LT>::chunk(ltid, p)) = r[p];
}

// ---------------------------------------------------------------- plain matrix loader, type C, bf16
// X[K][cols] row-major bf16 (reduction index slow).  ld, cols multiples of 8; chunks at or beyond `cols` read zeros.
template <int TILE, int LT = 256>
struct PlainCb {
  static constexpr int NV = StageMapCb<TILE, LT>::NV;
  struct Params { const void* p; int64_t ld; int cols; int K; };
  struct Raw { float4 v[NV]; };
  static constexpr bool kTypeR = false;
  const char* base;
  int64_t ldb;          // bytes
  uint32_t voff[NV];
  int K, kr;
  __device__ __forceinline__ void init(const Params& q, int col0, int tid, int /*ks0*/) {
    K = q.K; kr = StageMapCb<TILE, LT>::krow(tid); ldb = q.ld * 2;
    base = reinterpret_cast<const char*>(q.p) + (int64_t)col0 * 2;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int cl = 8 * StageMapCb<TILE, LT>::chunk(tid, p);
      voff[p] = col0 + cl < q.cols ? (uint32_t)(kr * (int)q.ld + cl) * 2u : BUF_OOB;
    }
  }
  __device__ __forceinline__ void issue(int ks, Raw& r) {
    const int rem = K - ks * BKB;
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(base + (int64_t)ks * BKB * ldb);
    const bool rowok = kr < rem;                      // rem >= 64 on full steps
#pragma unroll
    for (int p = 0; p < NV; ++p) r.v[p] = buf_load16(rs, rowok ? voff[p] : BUF_OOB);
  }
  __device__ __forceinline__ void finish(const Raw& r, float4 (&o)[NV]) const {
#pragma unroll
    for (int p = 0; p < NV; ++p) o[p] = r.v[p];
  }
};

// ---------------------------------------------------------------- loader role (as loader_loop, bf16 smem layout)
template <class Cfg, class L, bool IS_A>
__device__ __forceinline__ void stage_store_b(const L& ld, const typename L::Raw& raw, float* dst, int ltid) {
  constexpr int TILE = IS_A ? Cfg::BM : Cfg::BN;
  if constexpr (L::kTypeR) {
    constexpr int NV = IS_A ? Cfg::NVA : Cfg::NVB;
    float4 r[NV];
    ld.finish(raw, r);
    lds_store_R<TILE, NV, Cfg::LT>(dst, r, ltid);
  } else {
    constexpr int NV = StageMapCb<TILE, Cfg::LT>::NV;
    float4 r[NV];
    ld.finish(raw, r);
    lds_store_Cb<TILE, NV, Cfg::LT>(dst, r, ltid);
  }
}

template <class Cfg, class AL, class BL>
__device__ __forceinline__ void loader_loop_b(AL& al, BL& bl, int ks0, int ks1, float* smem) {
  constexpr int D = Cfg::PREFETCH;
  using SL = SmemLayoutB<Cfg, AL::kTypeR, BL::kTypeR>;
  const int ltid = loader_tid<Cfg>();
  float* const As0 = smem;
  float* const Bs0 = smem + 2 * SL::ABUF;
  typename AL::Raw rawA[D];
  typename BL::Raw rawB[D];
  al.issue(ks0, rawA[0]);
  bl.issue(ks0, rawB[0]);
  stage_store_b<Cfg, AL, true>(al, rawA[0], As0, ltid);
  stage_store_b<Cfg, BL, false>(bl, rawB[0], Bs0, ltid);
#pragma unroll
  for (int d = 0; d < D; ++d) {
    al.issue(ks0 + 1 + d, rawA[d]);
    bl.issue(ks0 + 1 + d, rawB[d]);
  }
  __syncthreads();
  for (int ks = ks0; ks < ks1; ks += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (ks + d < ks1) {
        const int nxt = ((ks + d - ks0) & 1) ^ 1;
        stage_store_b<Cfg, AL, true>(al, rawA[d], As0 + nxt * SL::ABUF, ltid);
        al.issue(ks + d + 1 + D, rawA[d]);
        stage_store_b<Cfg, BL, false>(bl, rawB[d], Bs0 + nxt * SL::BBUF, ltid);
        bl.issue(ks + d + 1 + D, rawB[d]);
        __syncthreads();
      }
    }
  }
}

// ---------------------------------------------------------------- MFMA role
// One K-step = 4 MFMA groups m = 0..3 of TM x TN v_mfma_f32_32x32x16_bf16; the fragments of group m+1 are fetched
// before the MFMAs of group m; the K-step barrier sits in front of the last group (as mfma_loop_eb).
template <class Cfg, bool AR, bool BR>
__device__ __forceinline__ void mfma_loop_b(f32x16 (&acc)[Cfg::TM][Cfg::TN], int ks0, int ks1, const float* smem) {
  using SL = SmemLayoutB<Cfg, AR, BR>;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const int l31 = lane & 31, h = lane >> 5, i16 = lane & 15, grp = (lane >> 4) & 1;
  constexpr int RSA = LdsImageB<Cfg::BM>::RSB, RSBb = LdsImageB<Cfg::BN>::RSB;
  // per-lane base offsets in BYTES from the start of an A / B stage
  const int a_off = AR ? ((wm * Cfg::WM + l31) * LDS_RS + 4 * h) * 4
                       : (8 * h + (i16 >> 2)) * RSA + (wm * Cfg::WM + 16 * grp + 4 * (i16 & 3)) * 2;
  const int b_off = BR ? ((wn * Cfg::WN + l31) * LDS_RS + 4 * h) * 4
                       : (8 * h + (i16 >> 2)) * RSBb + (wn * Cfg::WN + 16 * grp + 4 * (i16 & 3)) * 2;
  const char* const sm = reinterpret_cast<const char*>(smem);
  const char* const As0 = sm + a_off;
  const char* const Bs0 = sm + 2 * SL::ABUF * 4 + b_off;
  bf16x8 a[2][Cfg::TM], b[2][Cfg::TN];
  auto trread = [](const char* p) -> s16x4 {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  };
  auto fetch = [&](const char* ap, const char* bp, int m, int buf) {
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) {
      if (AR) {
        a[buf][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(ap + (32 * i * LDS_RS + 8 * m) * 4));
      } else {
        const char* q = ap + (16 * m) * RSA + 64 * i;           // 32 tile columns = 64 bytes
        const s16x4 lo = trread(q), hi = trread(q + 4 * RSA);
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        a[buf][i] = __builtin_bit_cast(bf16x8, v);
      }
    }
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      if (BR) {
        b[buf][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(bp + (32 * j * LDS_RS + 8 * m) * 4));
      } else {
        const char* q = bp + (16 * m) * RSBb + 64 * j;
        const s16x4 lo = trread(q), hi = trread(q + 4 * RSBb);
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        b[buf][j] = __builtin_bit_cast(bf16x8, v);
      }
    }
  };
  auto mma = [&](int buf) {
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[buf][i], b[buf][j], acc[i][j], 0, 0, 0);
  };
  __builtin_amdgcn_s_setprio(VQA_PRIO_MFMA);
  __syncthreads();
  fetch(As0, Bs0, 0, 0);
  for (int ks = ks0; ks < ks1; ++ks) {
    const int cur = (ks - ks0) & 1;
    const char* const Ac = As0 + cur * SL::ABUF * 4;
    const char* const Bc = Bs0 + cur * SL::BBUF * 4;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      fetch(Ac, Bc, m + 1, (m + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(m & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // every fragment of this stage is in registers; the next stage is complete
    if (ks + 1 < ks1) fetch(As0 + (cur ^ 1) * SL::ABUF * 4, Bs0 + (cur ^ 1) * SL::BBUF * 4, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma(1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Whole contraction over K-steps [ks0, ks1): true for MFMA waves (they hold the accumulators), false for loader waves.
template <class Cfg, class AL, class BL, class Init>
__device__ __forceinline__ bool gemm_mainloop_b(Init&& init, f32x16 (&acc)[Cfg::TM][Cfg::TN], int ks0, int ks1,
                                                float* smem) {
  if (is_loader_wave<Cfg>()) {
    AL al; BL bl;
    init(al, bl);
    loader_loop_b<Cfg>(al, bl, ks0, ks1, smem);
    return false;
  }
  mfma_loop_b<Cfg, AL::kTypeR, BL::kTypeR>(acc, ks0, ks1, smem);
  return true;
}

// Accumulators -> row-major bf16 matrix out[rows][ld] (elements); same addressing scheme as store_acc_tiles.
template <class Cfg>
__device__ __forceinline__ void store_acc_tiles_bf16(f32x16 (&acc)[Cfg::TM][Cfg::TN], uint16_t* out, int64_t ld, int rows,
                                                     int cols, int m0, int n0, int wm, int wn, int lane) {
  const bool interior = m0 + Cfg::BM <= rows && n0 + Cfg::BN <= cols;
  const uint32_t ldb = (uint32_t)ld * 2u;
  const uint32_t vl = (uint32_t)(4 * (lane >> 5)) * ldb + 2u * (uint32_t)(lane & 31);
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
          buf_store2(rs, bf16_bits(acc[i][j][r]), ok ? vl : BUF_OOB, (uint32_t)dr * ldb);
        }
      }
  };
  if (interior) body(std::true_type{}); else body(std::false_type{});
}

}  // namespace vqa
