// fp32 contraction on the bf16 matrix cores: the "3 x bf16" instantiation of the GEMM engine for gfx950.
//
// Every fp32 operand element is split EXACTLY into three bf16 terms  x = hi + mid + lo  (hi = RNE(x), mid = RNE(x - hi),
// lo = x - hi - mid: 8 + 8 + 8 significand bits cover fp32's 24), and a product a*b is accumulated in fp32 from the six
// partial products whose weight is at least 2^-16 of it:
//     a*b ~= a_hi*b_hi + a_hi*b_mid + a_mid*b_hi + a_mid*b_mid + a_hi*b_lo + a_lo*b_hi
// The three dropped terms (mid*lo, lo*mid, lo*lo) are below 2^-24 |a*b| each -- smaller than the rounding of ONE fp32
// product -- so the result carries fp32 accuracy (tests/test_x3_gpu.py measures it against float64 beside the native fp32
// MFMA path), while six v_mfma_f32_32x32x16_bf16 (32 cycles each) replace the eight v_mfma_f32_32x32x2_f32 (64 cycles
// each) of a 32x32x16 block: 2.67x the fp32 MFMA rate (2 500 / 6 = 416.7 TFLOP/s fp32-equivalent on MI355X).
//
// Structure: the fp32 engine's loaders unchanged (same Raw rings, same 32-element K-step, fp32 arg-max routing), the
// split done by the loader waves when a K-step goes to LDS, three bf16 planes per operand in LDS, the bf16 engine's
// fragment reads (ds_read_b128 for k-contiguous operands, ds_read_b64_tr_b16 for reduction-major ones).  A K-step stage
// of a (BM + BN)-row tile takes 240 bytes per row, so the tiles run one workgroup per CU (192 x 128: 153.6 KB) with
// fully double-buffered fragments (256 VGPRs per wave).
#pragma once
#include "bf16_core.hpp"

namespace vqa {

// type R image: per plane [row][XRS dwords] = 32 bf16 + 4 dwords of padding; three planes one after the other.  Row stride
// 20 dwords: the 16 rows of a ds_read_b128 pass start at 16 distinct multiples of 4 banks (conflict-free), and the four
// rows of a ds_write_b64 half-wave (8 lanes x 8 bytes per row) overlap in 12 of 64 banks only (three planes side by side
// in one 60-dword row overlapped in 36 of 64).
constexpr int XRS = 20;

template <int TILE>
struct LdsImageX {               // type C image: per plane [k 0..31][tile] of bf16, row stride = 64 / 192 (mod 256) bytes
  static constexpr int RAW = TILE * 2;
  static constexpr int RSB = RAW + ((RAW % 256 == 64 || RAW % 256 == 192) ? 0 : 64);
  static constexpr int PLANE = BK * RSB;            // bytes
  static constexpr int DWORDS = 3 * PLANE / 4;
};
template <class Cfg, bool AR, bool BR>
struct SmemLayoutX {
  static constexpr int ABUF = AR ? 3 * XRS * Cfg::BM : LdsImageX<Cfg::BM>::DWORDS;   // dwords per stage
  static constexpr int BBUF = BR ? 3 * XRS * Cfg::BN : LdsImageX<Cfg::BN>::DWORDS;
  static constexpr int BYTES = 2 * (ABUF + BBUF) * 4;
  static_assert(BYTES <= 160 * 1024, "tile does not fit the LDS");
};

// v = hi + mid + lo, four elements at a time; each output is two dwords of packed bf16 (element 0 in the low half).
// hi = RNE(v) (one v_cvt_pk_bf16_f32 per pair); the residual v - hi comes from v_dot2c_f32_bf16 with the constant pairs
// (-1, 0) / (0, -1): acc = v; acc += hi.lo * -1 + hi.hi * 0 -- exact (the products and the sum are representable), and one
// instruction per element instead of unpack + subtract.  3.5 VALU instructions per element in all.
// The constant pairs are handed over in registers the compiler cannot see through: as an inline constant, hipcc (ROCm 7.2)
// encodes the pair (-1, 0) as the operand "-1.0", which the hardware reads as the 32-bit pattern 0xbf800000 = (0, -1).
struct SplitConsts { uint32_t lo, hi; };
__device__ __forceinline__ SplitConsts split_consts() {
  SplitConsts c;
  asm volatile("s_mov_b32 %0, 0x0000bf80\n\ts_mov_b32 %1, 0xbf800000" : "=s"(c.lo), "=s"(c.hi));
  return c;
}
__device__ __forceinline__ float resid(uint32_t pk, uint32_t k, float x) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, pk), __builtin_bit_cast(bf2, k), x, false);
}
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& mid, uint2& lo, const SplitConsts k) {
#ifdef VQA_X3_EXP_NOSPLIT   // timing experiment only (tools/build_x3_variant.sh): no split arithmetic, results are garbage
  hi = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y)); mid = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w)); lo = hi;
  return;
#endif
  hi.x = pack_bf16x2(v.x, v.y);
  hi.y = pack_bf16x2(v.z, v.w);
#ifdef VQA_X3_SPLIT_SUB     // unpack + subtract instead of the dot2 residual (same values)
  const float r0 = v.x - bf16_lo(hi.x), r1 = v.y - bf16_hi(hi.x), r2 = v.z - bf16_lo(hi.y), r3 = v.w - bf16_hi(hi.y);
  mid.x = pack_bf16x2(r0, r1);
  mid.y = pack_bf16x2(r2, r3);
  lo.x = pack_bf16x2(r0 - bf16_lo(mid.x), r1 - bf16_hi(mid.x));
  lo.y = pack_bf16x2(r2 - bf16_lo(mid.y), r3 - bf16_hi(mid.y));
#else
  const float r0 = resid(hi.x, k.lo, v.x), r1 = resid(hi.x, k.hi, v.y), r2 = resid(hi.y, k.lo, v.z), r3 = resid(hi.y, k.hi, v.w);
  mid.x = pack_bf16x2(r0, r1);
  mid.y = pack_bf16x2(r2, r3);
  lo.x = pack_bf16x2(resid(mid.x, k.lo, r0), resid(mid.x, k.hi, r1));
  lo.y = pack_bf16x2(resid(mid.y, k.lo, r2), resid(mid.y, k.hi, r3));
#endif
}

// One value at a time (epilogues that write an activation in the x3-packed form): the same three terms as split4.
__device__ __forceinline__ void split1(float v, uint16_t& h, uint16_t& m, uint16_t& l) {
  h = bf16_bits(v);
  const float r1 = v - __uint_as_float((uint32_t)h << 16);
  m = bf16_bits(r1);
  l = bf16_bits(r1 - __uint_as_float((uint32_t)m << 16));
}
// x3-packed activations (include/vqa_hip.h vqa_x3_pack): byte offset of channel l31 inside its 32-channel group
__device__ __forceinline__ uint32_t x3p_lane(int l31) { return (uint32_t)((l31 >> 2) * 24 + (l31 & 3) * 2); }

// ---------------------------------------------------------------- operands split ahead of time
// A loader may deliver its chunks already split (weights: split once per step by vqa_x3_split instead of by every
// workgroup in every K-step).  Such a loader has kPreSplit = true and planes(raw, p, hi, mid, lo) instead of finish().
template <class L, class = void> struct is_presplit : std::false_type {};
template <class L> struct is_presplit<L, std::void_t<decltype(L::kPreSplit)>> : std::bool_constant<L::kPreSplit> {};

// Type C over a pre-split row-major matrix: three planes X_hi, X_mid, X_lo [K][cols] of bf16, `plane` elements apart.
// Thread -> (k row, 4-column chunks) as PlainC; a chunk is one 8-byte load per plane.  cols, ld multiples of 4.
template <int NV, int LT = 256>
struct PlainCx {
  struct Params { const void* p; int64_t ld; int cols; int K; int64_t plane; };
  struct Raw { uint2 v[NV][3]; };
  static constexpr bool kTypeR = false;
  static constexpr bool kPreSplit = true;
  const char* base;
  int64_t ldb, planeb;
  uint32_t voff[NV];
  int K, kr;
  __device__ __forceinline__ void init(const Params& q, int col0, int tid, int /*ks0*/) {
    K = q.K; kr = StageMap<LT>::c_krow(tid); ldb = q.ld * 2; planeb = q.plane * 2;
    base = reinterpret_cast<const char*>(q.p) + (int64_t)col0 * 2;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int cl = 4 * StageMap<LT>::c_chunk(tid, p);
      voff[p] = col0 + cl < q.cols ? (uint32_t)(kr * (int)q.ld + cl) * 2u : BUF_OOB;
    }
  }
  __device__ __forceinline__ void issue(int ks, Raw& r) {
    const bool rowok = kr < K - ks * BK;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const __amdgpu_buffer_rsrc_t rs = buf_rsrc(base + pl * planeb + (int64_t)ks * BK * ldb);
#pragma unroll
      for (int p = 0; p < NV; ++p) r.v[p][pl] = buf_load8(rs, rowok ? voff[p] : BUF_OOB);
    }
  }
  __device__ __forceinline__ void planes(const Raw& r, int p, uint2& hi, uint2& mid, uint2& lo) const {
    hi = r.v[p][0]; mid = r.v[p][1]; lo = r.v[p][2];
  }
};

// ---------------------------------------------------------------- loader role
template <class Cfg, class L, bool IS_A>
__device__ __forceinline__ void stage_store_x(const L& ld, const typename L::Raw& raw, float* dst, int ltid,
                                              const SplitConsts k) {
  constexpr int NV = IS_A ? Cfg::NVA : Cfg::NVB;
  constexpr int TILE = IS_A ? Cfg::BM : Cfg::BN;
  float4 r[NV];
  if constexpr (!is_presplit<L>::value) ld.finish(raw, r);
  char* const d = reinterpret_cast<char*>(dst);
#pragma unroll
  for (int p = 0; p < NV; ++p) {
    uint2 h, m, l;
    if constexpr (is_presplit<L>::value) ld.planes(raw, p, h, m, l); else split4(r[p], h, m, l, k);
#ifdef VQA_X3_EXP_NOWRITE   // timing experiment: loads + split, no LDS writes
    asm volatile("" ::"v"(h.x), "v"(h.y), "v"(m.x), "v"(m.y), "v"(l.x), "v"(l.y));
    continue;
#endif
    if constexpr (L::kTypeR) {
      char* q = d + (StageMap<Cfg::LT>::r_row(ltid, p) * XRS + 2 * StageMap<Cfg::LT>::r_chunk(ltid)) * 4;
      *reinterpret_cast<uint2*>(q) = h;
      *reinterpret_cast<uint2*>(q + TILE * XRS * 4) = m;
      *reinterpret_cast<uint2*>(q + 2 * TILE * XRS * 4) = l;
    } else {
      char* q = d + StageMap<Cfg::LT>::c_krow(ltid) * LdsImageX<TILE>::RSB + 8 * StageMap<Cfg::LT>::c_chunk(ltid, p);
      *reinterpret_cast<uint2*>(q) = h;
      *reinterpret_cast<uint2*>(q + LdsImageX<TILE>::PLANE) = m;
      *reinterpret_cast<uint2*>(q + 2 * LdsImageX<TILE>::PLANE) = l;
    }
  }
}

template <class Cfg, class AL, class BL>
__device__ __forceinline__ void loader_loop_x(AL& al, BL& bl, int ks0, int ks1, float* smem) {
  constexpr int D = Cfg::PREFETCH;
  using SL = SmemLayoutX<Cfg, AL::kTypeR, BL::kTypeR>;
  const int ltid = loader_tid<Cfg>();
  const SplitConsts kc = split_consts();
  float* const As0 = smem;
  float* const Bs0 = smem + 2 * SL::ABUF;
  typename AL::Raw rawA[D];
  typename BL::Raw rawB[D];
  al.issue(ks0, rawA[0]);
  bl.issue(ks0, rawB[0]);
  stage_store_x<Cfg, AL, true>(al, rawA[0], As0, ltid, kc);
  stage_store_x<Cfg, BL, false>(bl, rawB[0], Bs0, ltid, kc);
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
        stage_store_x<Cfg, AL, true>(al, rawA[d], As0 + nxt * SL::ABUF, ltid, kc);
#ifndef VQA_X3_EXP_NOLOAD   // timing experiment: the same registers are split and stored again, no global loads
        al.issue(ks + d + 1 + D, rawA[d]);
#endif
        stage_store_x<Cfg, BL, false>(bl, rawB[d], Bs0 + nxt * SL::BBUF, ltid, kc);
#ifndef VQA_X3_EXP_NOLOAD
        bl.issue(ks + d + 1 + D, rawB[d]);
#endif
        __syncthreads();
      }
    }
  }
}

// ---------------------------------------------------------------- MFMA role
// One K-step = 2 groups m = 0, 1 of 6 * TM * TN MFMAs; the fragments of the next group are fetched before the MFMAs of
// the current one; the K-step barrier sits in front of the last group (every fragment of the stage is in registers).
// Fragment reads of one group go in two parts, A and B: each part stays below the 15 operations lgkmcnt can count, so the
// wait in front of an MFMA block covers exactly the reads it consumes (with all 21 reads of a group issued at once the
// compiler had to wait for the 7 oldest of the NEW reads too: ~200 exposed cycles per group).
template <class Cfg, bool AR, bool BR>
struct MfmaX {
  using SL = SmemLayoutX<Cfg, AR, BR>;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  static constexpr int RSA = LdsImageX<Cfg::BM>::RSB, RSBb = LdsImageX<Cfg::BN>::RSB;
  static constexpr int PLA = AR ? Cfg::BM * XRS * 4 : LdsImageX<Cfg::BM>::PLANE;     // plane stride, bytes
  static constexpr int PLB = BR ? Cfg::BN * XRS * 4 : LdsImageX<Cfg::BN>::PLANE;
  const char* As0;
  const char* Bs0;
  bf16x8 a[2][3][Cfg::TM], b[2][3][Cfg::TN];

  __device__ __forceinline__ void setup(const float* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
    const int l31 = lane & 31, h = lane >> 5, i16 = lane & 15, grp = (lane >> 4) & 1;
    const int a_off = AR ? ((wm * Cfg::WM + l31) * XRS + 4 * h) * 4
                         : (8 * h + (i16 >> 2)) * RSA + (wm * Cfg::WM + 16 * grp + 4 * (i16 & 3)) * 2;
    const int b_off = BR ? ((wn * Cfg::WN + l31) * XRS + 4 * h) * 4
                         : (8 * h + (i16 >> 2)) * RSBb + (wn * Cfg::WN + 16 * grp + 4 * (i16 & 3)) * 2;
    const char* const sm = reinterpret_cast<const char*>(smem);
    As0 = sm + a_off;
    Bs0 = sm + 2 * SL::ABUF * 4 + b_off;
  }
  static __device__ __forceinline__ s16x4 trread(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  }
  __device__ __forceinline__ void fetchA(int stage, int m, int buf) {
    const char* ap = As0 + stage * SL::ABUF * 4;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) {
        if (AR) {
          a[buf][pl][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(ap + pl * PLA + 32 * i * XRS * 4 + 32 * m));
        } else {
          const char* q = ap + pl * PLA + (16 * m) * RSA + 64 * i;
          const s16x4 lo = trread(q), hi = trread(q + 4 * RSA);
          const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          a[buf][pl][i] = __builtin_bit_cast(bf16x8, v);
        }
      }
  }
  __device__ __forceinline__ void fetchB(int stage, int m, int buf) {
    const char* bp = Bs0 + stage * SL::BBUF * 4;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) {
        if (BR) {
          b[buf][pl][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(bp + pl * PLB + 32 * j * XRS * 4 + 32 * m));
        } else {
          const char* q = bp + pl * PLB + (16 * m) * RSBb + 64 * j;
          const s16x4 lo = trread(q), hi = trread(q + 4 * RSBb);
          const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          b[buf][pl][j] = __builtin_bit_cast(bf16x8, v);
        }
      }
  }
  // plane pairs, small terms first: (lo,hi) (hi,lo) (mid,mid) | (mid,hi) (hi,mid) (hi,hi); consecutive MFMAs go to
  // different accumulators
  template <int H>
  __device__ __forceinline__ void mma(f32x16 (&acc)[Cfg::TM][Cfg::TN], int buf) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#ifdef VQA_X3_EXP_NPROD     // timing experiment: only the last NPROD partial products
    constexpr int T0 = 6 - VQA_X3_EXP_NPROD;
#else
    constexpr int T0 = 0;
#endif
#pragma unroll
    for (int t = 3 * H; t < 3 * H + 3; ++t) {
      if (t < T0) continue;
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[buf][PA[t]][i], b[buf][PB[t]][j], acc[i][j], 0, 0, 0);
    }
  }
  // first fragments of a stage (group 0 into buffer 0): before the first K-step of a run
  __device__ __forceinline__ void prime(int stage) {
    fetchA(stage, 0, 0);
    fetchB(stage, 0, 0);
  }
  // one K-step on stage `cur` (its group 0 already in buffer 0); has_next: the other stage holds a further K-step
  __device__ __forceinline__ void kstep(f32x16 (&acc)[Cfg::TM][Cfg::TN], int cur, bool has_next) {
    fetchA(cur, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma<0>(acc, 0);
    __builtin_amdgcn_sched_barrier(0);
    fetchB(cur, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma<1>(acc, 0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();   // every fragment of this stage is in registers; the next stage is complete
    if (has_next) fetchA(cur ^ 1, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma<0>(acc, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) fetchB(cur ^ 1, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma<1>(acc, 1);
    __builtin_amdgcn_sched_barrier(0);
  }
};

template <class Cfg, bool AR, bool BR>
__device__ __forceinline__ void mfma_loop_x(f32x16 (&acc)[Cfg::TM][Cfg::TN], int ks0, int ks1, const float* smem) {
  MfmaX<Cfg, AR, BR> mx;
  mx.setup(smem);
  __builtin_amdgcn_s_setprio(VQA_PRIO_MFMA);
  __syncthreads();
  mx.prime(0);
  for (int ks = ks0; ks < ks1; ++ks) mx.kstep(acc, (ks - ks0) & 1, ks + 1 < ks1);
}

// Whole contraction over K-steps [ks0, ks1): true for MFMA waves (they hold the accumulators), false for loader waves.
template <class Cfg, class AL, class BL, class Init>
__device__ __forceinline__ bool gemm_mainloop_x(Init&& init, f32x16 (&acc)[Cfg::TM][Cfg::TN], int ks0, int ks1,
                                                float* smem) {
  if (is_loader_wave<Cfg>()) {
    AL al; BL bl;
    init(al, bl);
    loader_loop_x<Cfg>(al, bl, ks0, ks1, smem);
    return false;
  }
  mfma_loop_x<Cfg, AL::kTypeR, BL::kTypeR>(acc, ks0, ks1, smem);
  return true;
}

// ---------------------------------------------------------------- persistent tiles
// As gemm_persistent (gemm_core.hpp): a workgroup walks the tiles first, first + stride, ... (< ntiles), every tile the
// same K-steps [0, nk); the loader waves treat them as one flat sequence of K-steps, so the first stages of tile t+1 are
// loaded, split and in LDS while the MFMA waves run the epilogue of tile t.  With ONE workgroup per CU there is no second
// workgroup to cover a tile's prologue and epilogue, which is what this buys back.
//   init_tile(tile, AL&, BL&)  : (re)initialise both loaders for a tile;   epilogue(tile, acc) : MFMA waves, no barriers
template <class Cfg, class AL, class BL, class InitTile, class Epilogue>
__device__ __forceinline__ void gemm_persistent_x(int first, int stride, int ntiles, int nk, float* smem,
                                                  InitTile&& init_tile, Epilogue&& epilogue) {
  using SL = SmemLayoutX<Cfg, AL::kTypeR, BL::kTypeR>;
  constexpr int D = Cfg::PREFETCH;
  if (first >= ntiles) return;
  const int my_tiles = (ntiles - first + stride - 1) / stride;
  const int total = my_tiles * nk;
  if (is_loader_wave<Cfg>()) {
    const int ltid = loader_tid<Cfg>();
    const SplitConsts kc = split_consts();
    float* const As0 = smem;
    float* const Bs0 = smem + 2 * SL::ABUF;
    AL al; BL bl;
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
    stage_store_x<Cfg, AL, true>(al, rawA[0], As0, ltid, kc);
    stage_store_x<Cfg, BL, false>(bl, rawB[0], Bs0, ltid, kc);
#pragma unroll
    for (int d = 0; d < D; ++d) next(rawA[d], rawB[d]);
    __syncthreads();
    for (int s = 0; s < total; s += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (s + d < total) {
          const int nxt = ((s + d) & 1) ^ 1;
          stage_store_x<Cfg, AL, true>(al, rawA[d], As0 + nxt * SL::ABUF, ltid, kc);
          stage_store_x<Cfg, BL, false>(bl, rawB[d], Bs0 + nxt * SL::BBUF, ltid, kc);
          next(rawA[d], rawB[d]);
          __syncthreads();
        }
      }
    }
    return;
  }
  MfmaX<Cfg, AL::kTypeR, BL::kTypeR> mx;
  mx.setup(smem);
  __builtin_amdgcn_s_setprio(VQA_PRIO_MFMA);
  __syncthreads();
  mx.prime(0);
  int s = 0;
  for (int tile = first; tile < ntiles; tile += stride) {
    f32x16 acc[Cfg::TM][Cfg::TN];
    acc_zero<Cfg>(acc);
    for (int ks = 0; ks < nk; ++ks, ++s) mx.kstep(acc, s & 1, s + 1 < total);
    epilogue(tile, acc);
  }
}

}  // namespace vqa
