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
//     conv_patch_bf16.hip) one stage ahead, ACROSS tile boundaries, six pieces per wave interleaved with its 16 MFMAs;
//     one counted s_waitcnt + one s_barrier per stage;
//   * LDS image: rows of 128 bytes, 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7) (applied to the DMA's per-lane
//     source address and to the fragment read): the 16 rows of a ds_read_b128 lane group then cover all 16 bank groups;
//   * epilogue per tile: row-group term (two groups at most per tile: rg_div >= 256), ReLU, bf16, through a wave-private LDS
//     scratch, 16 bytes per lane to HBM (whole 128-byte runs).
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
constexpr int TG_SCR = 32 * 128;                  // epilogue scratch per wave: 32 rows x 64 bf16
constexpr int TG_LDS = 2 * TG_STAGE + 8 * TG_SCR; // 128 KiB

struct TgParams {
  const char* A; const char* A_end; int64_t lda;   // bf16 [M][K], lda in elements
  const char* W; int64_t ldw;                      // bf16 [N][K]
  uint16_t* C; int64_t ldc;                        // bf16 [M][N]
  const float* rg; int64_t rg_ld; int rg_div; int rg_op;   // optional row-group term (add / mul), groups of rg_div rows
  int relu;
  int M, N, K, tiles_m, tiles_n, nk;
};

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

  tg_rsrc_t ns_a = tg_rsrc(P.A), ns_w = tg_rsrc(P.W);
  uint32_t ns_k = 0;
  int ns_buf = 0;
  bool ns_on = false;
  auto next_stage = [&](int tile, int ks, int buf, bool on) {
    const int mt = tile / P.tiles_n, nt = tile - mt * P.tiles_n;
    const char* a = P.A + (int64_t)mt * TG_BM * P.lda * 2;
    const int64_t left = P.A_end - a;                       // rows past M read zeros (range check)
    ns_a = tg_rsrc(a, left > 0xffff0000LL ? 0xffff0000u : (uint32_t)left);
    ns_w = tg_rsrc(P.W + (int64_t)nt * TG_BN * P.ldw * 2);
    ns_k = (uint32_t)(ks * TG_BK * 2);
    ns_buf = buf;
    ns_on = on;
  };
  auto issue_piece = [&](int q) {
    if (!ns_on) return;
    const int p = wave + 8 * q;
    if (p < 32) tg_dma16(ns_a, smem + ns_buf * TG_STAGE + p * 1024, voff[q], ns_k);
    else tg_dma16(ns_w, smem + ns_buf * TG_STAGE + TG_A + (p - 32) * 1024, voff[q], ns_k);
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
  char* const scr = smem + 2 * TG_STAGE + wave * TG_SCR;

  next_stage(first, 0, 0, true);
#pragma unroll
  for (int q = 0; q < 6; ++q) issue_piece(q);
  int buf = 0;
  bool after_epilogue = false;
  for (int t = 0; t < my_tiles; ++t) {
    const int tile = first + t * (int)gridDim.x;
    const int mt = tile / P.tiles_n, nt = tile - mt * P.tiles_n;
    const int m0 = mt * TG_BM, n0 = nt * TG_BN;
    // row-group terms of this tile (at most two groups: rg_div >= 256), loaded early, used in the epilogue
    float rgv[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    const int g0 = P.rg ? m0 / P.rg_div : 0;
    const int boundary = (g0 + 1) * P.rg_div;
    if (P.rg) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + 32 * j + r;
        rgv[0][j] = P.rg[(int64_t)g0 * P.rg_ld + col];
        rgv[1][j] = boundary < P.M ? P.rg[(int64_t)(g0 + 1) * P.rg_ld + col] : rgv[0][j];
      }
    }
    for (int ks = 0; ks < P.nk; ++ks) {
      // this wave's pieces of the current stage have landed (issued BEFORE the epilogue's 8 stores, if one came in between;
      // the row-group loads above are older than nothing that matters: their wait is hipcc's, at first use)
      if (after_epilogue) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      after_epilogue = false;
      __builtin_amdgcn_s_barrier();
      {
        int nks = ks + 1, ntile = tile;
        bool more = true;
        if (nks == P.nk) { nks = 0; ntile = tile + (int)gridDim.x; more = t + 1 < my_tiles; }
        next_stage(ntile, nks, buf ^ 1, more);
      }
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
      buf ^= 1;
    }
    // ---- epilogue of the tile: (+|*) row-group term, ReLU, bf16, through the wave's scratch in two halves of 32 rows
    const bool mul = P.rg_op != 0, relu = P.relu != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row0 = m0 + wm * 64 + 32 * i + 4 * h;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int dr = (e & 3) + 8 * (e >> 2);
          float v = acc[i][j][e];
          if (P.rg) {
            const float tt = row0 + dr < boundary ? rgv[0][j] : rgv[1][j];
            v = mul ? v * tt : v + tt;
          }
          if (relu) v = fmaxf(v, 0.f);
          *reinterpret_cast<uint16_t*>(scr + (4 * h + dr) * 128 + (32 * j + r) * 2) = bf16_bits(v);
        }
      asm volatile("" ::: "memory");
      const int rbase = m0 + wm * 64 + 32 * i;
      const __amdgpu_buffer_rsrc_t ro = buf_rsrc(P.C + (int64_t)rbase * P.ldc + n0 + wn * 64);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int byte = q * 1024 + lane * 16;
        const int row = byte >> 7, inrow = byte & 127;
        const float4 v = *reinterpret_cast<const float4*>(scr + byte);
        const bool ok = rbase + row < P.M;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro,
                                               ok ? (int)((uint32_t)row * (uint32_t)P.ldc * 2u + inrow) : (int)BUF_OOB, 0, 0);
      }
      asm volatile("" ::: "memory");
    }
    after_epilogue = true;
    zero_acc();
  }
}

}  // namespace vqa

using namespace vqa;

extern "C" {

/* 1 when vqa_gemm_tall_bf16 takes the shape (otherwise vqa_gemm_bf16): bf16 output, A [M][K] and W [N][K] k-contiguous,
 * K % 64 == 0, N % 128 == 0, at least 64 tiles of 256 x 128, row groups (if any) of at least 256 rows. */
int vqa_gemm_tall_bf16_supported(int M, int N, int K, int rg_div, int has_rowgroup) {
  if (M < 256 || N % 128 || K % 64 || K <= 0 || K > 4096) return 0;
  if ((int64_t)((M + 255) / 256) * (N / 128) < 64) return 0;
  if (has_rowgroup && rg_div < 256) return 0;
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
