// Tall bf16 GEMM with a SHORT reduction for gfx950: C[M][N] = act(A[M][K] . W[N][K]^T (+|*) rowgroup), bf16 result.
//
// The case: the attention stage's v_conv forward of the bf16 path (models/model.py:173,187-193) -- M = B * positions
// (1.5 M rows at 448 x 448, B = 512), N = 1024, K = 256, 3 GB of x = relu(v' + q') written per step.  With K = 256 a
// 128 x 128 tile lives for four K-steps: in the role-split engine (bf16.hip) a workgroup's prologue, epilogue and dispatch
// were > 90 % of its life (2.2-2.7 ms per launch, 14 % of the bf16 peak, a third of the HBM write rate).  Here:
//   * persistent workgroups (one per CU) walk 256 x 128 tiles, N fastest, so the eight column tiles of a row block follow
//     each other and its A rows come from L2 after the first;
//   * 8 waves, all computing: 4 (M) x 2 (N) waves of 64 x 64 = 2 x 2 accumulators of v_mfma_f32_32x32x16_bf16;
//   * a stage = 64 of K: A 256 rows x 128 B + W 128 rows x 128 B = 48 pieces of 1 KiB, fetched by LDS-DMA (inline asm, see
//     conv_patch_bf16.hip) TWO stages ahead (three stage buffers), ACROSS tile boundaries, six pieces per wave interleaved
//     with its 16 MFMAs; one counted s_waitcnt + one s_barrier per stage;
//   * LDS image: rows of 128 bytes, 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7) (applied to the DMA's per-lane
//     source address and to the fragment read): the 16 rows of a ds_read_b128 lane group then cover all 16 bank groups;
//   * epilogue per tile: row-group term (two groups at most per tile: rg_div >= 256; the tile's terms arrive as one more DMA
//     piece in a 1-KiB LDS table), ReLU, bf16, through a wave-private LDS scratch, 16 bytes per lane to HBM (whole 128-byte runs).
// Arithmetic intensity bounds this shape below the matrix peak (128 flop per byte of A): ~50 % is the ceiling.
#include "bf16_core.hpp"

namespace vqa {

typedef unsigned int tg_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ tg_rsrc_t tg_rsrc(const void* base, uint32_t bytes = 0xffff0000u) {
  const uint64_t a = (uint64_t)base;
  tg_rsrc_t r;
  r.x = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r.y = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
// 16 bytes per lane, global -> LDS without a VGPR round trip; invisible to hipcc's wait insertion (the kernel counts)
__device__ __forceinline__ void tg_dma16(tg_rsrc_t r, const void* lds_dst, uint32_t voff, uint32_t soff) {
  const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds_dst);
  uint32_t keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(m0v), "v"(voff), "s"(r), "s"(soff) : "memory");
}

constexpr int TG_BM = 256, TG_BN = 128, TG_BK = 64;
constexpr int TG_A = TG_BM * 128;                 // bytes of an A stage (256 rows x 64 bf16)
constexpr int TG_B = TG_BN * 128;
constexpr int TG_STAGE = TG_A + TG_B;             // 48 KiB
constexpr int TG_NST = 3;                         // stages in LDS: two in flight while one is computed
constexpr int TG_SCR = 8 * 128;                   // epilogue scratch per wave: 8 rows x 64 bf16
constexpr int TG_TAB = 1024;                      // row-group terms of the tile: [2 groups][128 columns] fp32
constexpr int TG_LDS = TG_NST * TG_STAGE + 8 * TG_SCR + TG_TAB;   // 153 KiB

struct TgParams {
  const char* A; const char* A_end; int64_t lda;   // bf16 [M][K], lda in elements
  const char* W; int64_t ldw;                      // bf16 [N][K]
  uint16_t* C; int64_t ldc;                        // bf16 [M][N]
  const float* rg; int64_t rg_ld; int rg_div; int rg_op; int rg_groups;   // optional row-group term (add / mul), groups of rg_div rows
  int relu;
  int M, N, K, tiles_m, tiles_n, nk;
};

// Stage ring.  With two stages (round-3 first version) the pieces of stage g + 1 were issued DURING stage g and waited for at its
// end: half a stage (~0.3 us) of lead against ~1-2 us of HBM latency -- every stage ended in a stall, 1.96 ms per launch where
// the HBM time is 0.9 ms.  Three stages: the pieces of stage g + 2 are issued during stage g; the wait at the top of stage
// g + 1 is COUNTED (the six pieces of stage g + 2 this wave issued last may still be in flight).  Every wave issues exactly six
// pieces per stage (out-of-range ones past the last tile: the hardware range check drops them), so the counts are constants;
// wave 0 adds the 1-KiB row-group table of a tile (nk >= 3 with row groups).
__global__ __launch_bounds__(512, 2) void gemm_tall_bf16_kernel(const TgParams P) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int ntiles = P.tiles_m * P.tiles_n;
  const int first = xcd_swizzle(blockIdx.x, gridDim.x);
  const int my_tiles = first < ntiles ? (ntiles - first + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  if (my_tiles == 0) return;
  const int nstages = my_tiles * P.nk;

  // ---- fragment read addresses: A rows wm*64 + 32 i + r, W rows wn*64 + 32 j + r; chunk 2 s + h of k-step s
  uint32_t aoff[2][4], boff[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ra = wm * 64 + 32 * i + r, rb = wn * 64 + 32 * i + r;
      aoff[i][s] = (uint32_t)(ra * 128 + (((2 * s + h) ^ ((ra >> 1) & 7)) << 4));
      boff[i][s] = (uint32_t)(TG_A + rb * 128 + (((2 * s + h) ^ ((rb >> 1) & 7)) << 4));
    }
  // ---- DMA pieces of this wave: piece p = wave + 8 q (q = 0..5): p < 32 -> A rows 8p.., else W rows 8(p-32)..
  uint32_t voff[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int p = wave + 8 * q;
    const int row = (p < 32 ? 8 * p : 8 * (p - 32)) + (lane >> 3), slot = lane & 7;
    const int chunk = slot ^ ((row >> 1) & 7);
    voff[q] = (uint32_t)(row * (p < 32 ? (int)P.lda : (int)P.ldw) * 2 + chunk * 16);
  }
  char* const tab = smem + TG_NST * TG_STAGE + 8 * TG_SCR;

  // the stage being fetched: two ahead of the one being computed
  tg_rsrc_t ns_a = tg_rsrc(P.A, 0u), ns_w = tg_rsrc(P.W, 0u);
  uint32_t ns_k = 0;
  int ns_buf = 0;
  int ft = 0, fks = 0, fcount = 0;                  // fetch cursor: tile number (of mine), k-stage, stages fetched so far
  auto set_fetch = [&]() {
    const bool on = fcount < nstages;
    const int tile = first + ft * (int)gridDim.x;
    const int mt = tile / P.tiles_n, nt = tile - mt * P.tiles_n;
    const char* a = P.A + (int64_t)mt * TG_BM * P.lda * 2;
    const int64_t left = P.A_end - a;                       // rows past M read zeros (range check)
    ns_a = tg_rsrc(a, !on || left <= 0 ? 0u : left > 0xffff0000LL ? 0xffff0000u : (uint32_t)left);
    ns_w = tg_rsrc(P.W + (int64_t)nt * TG_BN * P.ldw * 2, on ? 0xffff0000u : 0u);
    ns_k = (uint32_t)(fks * TG_BK * 2);
    ns_buf = fcount % TG_NST;
  };
  auto advance_fetch = [&]() {
    ++fcount;
    if (++fks == P.nk) { fks = 0; ++ft; }
  };
  auto issue_piece = [&](int q) {          // always issued: the counted waits rely on six per wave and stage
    const int p = wave + 8 * q;
    if (p < 32) tg_dma16(ns_a, smem + ns_buf * TG_STAGE + p * 1024, voff[q], ns_k);
    else tg_dma16(ns_w, smem + ns_buf * TG_STAGE + TG_A + (p - 32) * 1024, voff[q], ns_k);
  };
  // row-group table of a tile (wave 0): lanes 0-31 the tile's first group, 32-63 the next one, 4 columns each
  auto issue_table = [&](int tile) {
    const int mt = tile / P.tiles_n, nt = tile - mt * P.tiles_n;
    const int g0 = (mt * TG_BM) / P.rg_div + (lane >> 5);
    const tg_rsrc_t rr = tg_rsrc(P.rg, P.rg ? 0xffff0000u : 0u);
    const uint32_t vo = g0 < P.rg_groups ? (uint32_t)(((int64_t)g0 * P.rg_ld + nt * TG_BN + 4 * (lane & 31)) * 4) : 0xffff0000u;
    tg_dma16(rr, tab, vo, 0u);
  };

  f32x16 acc[2][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  };
  zero_acc();
  char* const scr = smem + TG_NST * TG_STAGE + wave * TG_SCR;

  // prologue: stages 0 and 1 in flight
#pragma unroll
  for (int pre = 0; pre < 2; ++pre) {
    set_fetch();
#pragma unroll
    for (int q = 0; q < 6; ++q) issue_piece(q);
    advance_fetch();
  }
  int buf = 0;
  // VMEM operations this wave issued AFTER the pieces of the stage it waits for next (they may stay in flight): wait_n; the
  // same for the stage after it: younger.  Updated as pieces, the table and the epilogue stores go out (all wave-uniform).
  int wait_n = 6, younger = 0;
  for (int t = 0; t < my_tiles; ++t) {
    const int tile = first + t * (int)gridDim.x;
    const int mt = tile / P.tiles_n, nt = tile - mt * P.tiles_n;
    const int m0 = mt * TG_BM, n0 = nt * TG_BN;
    for (int ks = 0; ks < P.nk; ++ks) {
      // this wave's pieces of the current stage have landed (s_waitcnt takes an immediate: the counts that occur are 6 / 7 between
      // stages, + 8 across an epilogue; anything else waits for everything)
      switch (wait_n) {
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
      __builtin_amdgcn_s_barrier();
      // the tile's row-group table goes out BEFORE the pieces issued in this stage (those of the tile's last stage when
      // ks == nk - 3): the wait at the top of that last stage then covers it, the barrier behind the wait shows it to everybody;
      // its previous reader, the previous tile's epilogue, is behind every wave that passed this barrier
      if (wave == 0 && P.rg && ks == P.nk - 3) { issue_table(tile); ++younger; }
      set_fetch();
      const char* const st = smem + buf * TG_STAGE;
      bf16x8 a[2], b[2];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i] = *reinterpret_cast<const bf16x8*>(st + aoff[i][s]);
          b[i] = *reinterpret_cast<const bf16x8*>(st + boff[i][s]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        issue_piece(s);
        if (s >= 2) issue_piece(s + 2);
        __builtin_amdgcn_sched_barrier(0);
      }
      advance_fetch();
      wait_n = younger + 6;          // the next stage's pieces are older than this stage's table piece and the six just issued
      younger = 0;
      buf = buf + 1 == TG_NST ? 0 : buf + 1;
    }
    // ---- epilogue of the tile: (+|*) row-group term, ReLU, bf16, through the wave's scratch in rounds of 8 rows
    const bool mul = P.rg_op != 0, relu = P.relu != 0;
    float rgv[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    const int g0 = P.rg ? m0 / P.rg_div : 0;
    const int boundary = (g0 + 1) * P.rg_div;
    if (P.rg) {
#pragma unroll
      for (int gg = 0; gg < 2; ++gg)
#pragma unroll
        for (int j = 0; j < 2; ++j) rgv[gg][j] = *reinterpret_cast<const float*>(tab + (gg * 128 + wn * 64 + 32 * j + r) * 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int rbase = m0 + wm * 64 + 32 * i + 8 * q4;         // the round's 8 rows: rbase + 4 h + (e & 3)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            float v = acc[i][j][4 * q4 + e4];
            if (P.rg) {
              const float tt = rbase + 4 * h + e4 < boundary ? rgv[0][j] : rgv[1][j];
              v = mul ? v * tt : v + tt;
            }
            if (relu) v = fmaxf(v, 0.f);
            *reinterpret_cast<uint16_t*>(scr + (4 * h + e4) * 128 + (32 * j + r) * 2) = bf16_bits(v);
          }
        asm volatile("" ::: "memory");
        const __amdgpu_buffer_rsrc_t ro = buf_rsrc(P.C + (int64_t)rbase * P.ldc + n0 + wn * 64);
        {
          const int row = lane >> 3, inrow = (lane & 7) * 16;
          const float4 v = *reinterpret_cast<const float4*>(scr + lane * 16);
          const bool ok = rbase + row < P.M;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro,
                                                 ok ? (int)((uint32_t)row * (uint32_t)P.ldc * 2u + inrow) : (int)BUF_OOB, 0, 0);
        }
        asm volatile("" ::: "memory");
      }
    }
    wait_n += 8;                     // the eight stores: younger than the pieces of both stages in flight
    younger += 8;
    zero_acc();
  }
}

}  // namespace vqa

using namespace vqa;

extern "C" {

/* 1 when vqa_gemm_tall_bf16 takes the shape (otherwise vqa_gemm_bf16): bf16 output, A [M][K] and W [N][K] k-contiguous,
 * K % 64 == 0, N % 128 == 0, at least 64 tiles of 256 x 128, row groups (if any) of at least 256 rows and K >= 192. */
int vqa_gemm_tall_bf16_supported(int M, int N, int K, int rg_div, int has_rowgroup) {
  if (M < 256 || N % 128 || K % 64 || K <= 0 || K > 4096) return 0;
  if ((int64_t)((M + 255) / 256) * (N / 128) < 64) return 0;
  if (has_rowgroup && (rg_div < 256 || K < 192)) return 0;     // the row-group table rides three stages ahead of its reader
  return 1;
}

int vqa_gemm_tall_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int M, int N, int K,
                       const float* rowgroup, int64_t rg_ld, int rg_div, int rg_op, int relu, int tag, vqa_stream_t stream) {
  VQA_REQUIRE(A && W && C, "vqa_gemm_tall_bf16: null operand");
  VQA_REQUIRE(vqa_gemm_tall_bf16_supported(M, N, K, rg_div, rowgroup != nullptr), "vqa_gemm_tall_bf16: unsupported shape %dx%dx%d", M, N, K);
  VQA_REQUIRE(lda >= K && ldw >= K && ldc >= N && lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && lda < (1 << 20) && ldw < (1 << 20),
              "vqa_gemm_tall_bf16: leading dimensions must be multiples of 8 (lda=%lld ldw=%lld ldc=%lld)", (long long)lda,
              (long long)ldw, (long long)ldc);
  VQA_REQUIRE((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(C) & 15) == 0, "vqa_gemm_tall_bf16: operands must be 16-byte aligned");
  TgParams P{};
  P.A = static_cast<const char*>(A); P.A_end = P.A + (int64_t)M * lda * 2; P.lda = lda;
  P.W = static_cast<const char*>(W); P.ldw = ldw;
  P.C = static_cast<uint16_t*>(C); P.ldc = ldc;
  P.rg = rowgroup; P.rg_ld = rg_ld; P.rg_div = rg_div > 0 ? rg_div : 1; P.rg_op = rg_op; P.relu = relu;
  P.rg_groups = (M + P.rg_div - 1) / P.rg_div;
  P.M = M; P.N = N; P.K = K;
  P.tiles_m = (M + TG_BM - 1) / TG_BM; P.tiles_n = N / TG_BN; P.nk = K / TG_BK;
  hipStream_t s = (hipStream_t)stream;
  set_launch_tag(tag);
  ProfScope prof(VQA_K_GEMM, s);
  int rc = ensure_dyn_smem(reinterpret_cast<const void*>(gemm_tall_bf16_kernel), TG_LDS, "attr(gemm_tall_bf16)");
  if (rc) return rc;
  const int tiles = P.tiles_m * P.tiles_n;
  hipLaunchKernelGGL(gemm_tall_bf16_kernel, dim3(tiles < 256 ? tiles : 256), dim3(512), TG_LDS, s, P);
  return check_hip(hipGetLastError(), "gemm_tall_bf16 launch");
}

}  // extern "C"
