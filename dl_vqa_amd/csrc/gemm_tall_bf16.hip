// Tall bf16 GEMM with a SHORT reduction for gfx950: C[M][N] = act(A[M][K] . W[N][K]^T (+|*) rowgroup), bf16 result.
//
// The case: the attention stage's v_conv forward of the bf16 path (models/model.py:173,187-193) -- M = B * positions
// (1.5 M rows at 448 x 448, B = 512), N = 1024, K = 256, 3 GB of x = relu(v' + q') written per step.  With K = 256 a
// 128 x 128 tile lives for four K-steps: in the role-split engine (bf16.hip) a workgroup's prologue, epilogue and dispatch
// were > 90 % of its life (2.2-2.7 ms per launch, 14 % of the bf16 peak, a third of the HBM write rate).  Here:
//   * persistent workgroups (one per CU) walk 256 x 128 tiles, N fastest, so the eight column tiles of a row block follow
//     each other and its A rows come from L2 after the first;
//   * 8 waves, all computing: 4 (M) x 2 (N) waves of 64 x 64 = 2 x 2 accumulators of v_mfma_f32_32x32x16_bf16;
//   * a stage = 64 of K: A 256 rows x 128 B + W 128 rows x 128 B = 48 pieces of 1 KiB, six per wave, fetched TWO stages ahead
//     through registers (see the note at the kernel), ACROSS tile boundaries; one s_barrier per stage;
//   * LDS image: rows of 128 bytes, 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7) (applied where the piece is stored
//     and to the fragment read): the 16 rows of a ds_read_b128 lane group then cover all 16 bank groups;
//   * epilogue per tile: row-group term (two groups at most per tile: rg_div >= 256; the tile's terms travel like one more
//     piece into a 1-KiB LDS table), ReLU, bf16, through a wave-private LDS scratch, 16 bytes per lane to HBM (whole 128-byte runs).
// Arithmetic intensity bounds this shape below the matrix peak (128 flop per byte of A): ~50 % is the ceiling.
#include "bf16_core.hpp"
#include <stdlib.h>

namespace vqa {

// Timing experiments (tools/kbench_tall.py --dbg ...): only a -DVQA_TALL_DIAG build looks at VQA_TALL_DBG (1 = no epilogue,
// 2 = no stage traffic after the prologue, 4 = no MFMA, 8 = every tile's stores go to tile 0's rows (the write stream stays in
// L2), 16 = the epilogue's arithmetic without its stores); the shipped kernel carries none of those branches.
#ifdef VQA_TALL_DIAG
#define TG_DBG(bit) (P.dbg & (bit))
#else
#define TG_DBG(bit) false
#endif

constexpr int TG_BN = 128, TG_BK = 64;
constexpr int TG_B = TG_BN * 128;
constexpr int TG_TAB = 1024;                      // row-group terms of the tile: [2 groups][128 columns] fp32
// WMW waves along M (64 rows each) x 2 along N: 4 -> 256 x 128 tiles, one workgroup of 8 waves per CU; 2 -> 128 x 128 tiles,
// TWO workgroups of 4 waves per CU.  Measured with the parts switched off (tools/kbench_tall.py, 1.49 M x 1024 x 256): barriers +
// fragment reads 0.31 ms, + MFMAs 0.62, stage traffic alone 0.45, epilogue arithmetic 0.2, its stores 0.1 into L2 and 0.5 into
// HBM -- and the times ADD (1.7 ms all together) for either workgroup shape, staggered starts included: what a tile costs is
// the sum of its phases' latencies, not the busiest unit.  DESIGN.md section 4.3 has the table.
template <int WMW>
struct TgCfg {
  static constexpr int BM = 64 * WMW, NW = 2 * WMW;
  static constexpr int A = BM * 128;              // bytes of an A stage (BM rows x 64 bf16)
  static constexpr int STAGE = A + TG_B;
  static constexpr int PA = BM / 8;               // 1-KiB pieces of the A stage; 16 of the W stage
  static constexpr int NP = (PA + 16) / NW;       // pieces per wave: 6 / 8, the first 4 of them A
  static_assert(PA == 4 * NW, "piece q of a wave is an A piece for q < 4");
  static constexpr int LDS = 2 * STAGE + TG_TAB;                 // 97 / 65 KiB
};

struct TgParams {
  const char* A; const char* A_end; int64_t lda;   // bf16 [M][K], lda in elements
  const char* W; int64_t ldw;                      // bf16 [N][K]
  uint16_t* C; int64_t ldc;                        // bf16 [M][N]
  const float* rg; int64_t rg_ld; int rg_div; int rg_op; int rg_groups;   // optional row-group term (add / mul), groups of rg_div rows
  int relu;
  int M, N, K, tiles_m, tiles_n, nk;
  int dbg;
};

// How the stages reach LDS.  A stage is 48 pieces of 1 KiB for 128 MFMAs: far more bytes per MFMA than the convolutions move,
// and the LDS-DMA form (buffer_load ... lds, one piece per instruction) is slow to ISSUE -- a few hundred cycles per instruction
// (conv_patch_bf16.hip hides nine of them behind 72 MFMAs per wave; here six stood against 16 MFMAs): with DMA, two or three
// stage buffers alike, the kernel sat at 1.7-2.0 ms per launch, 20 k cycles per tile where MFMAs + epilogue are ~8 k, with HBM
// traffic at the compulsory 3.8 GB (PMC).  So the pieces travel through REGISTERS: a wave loads its six pieces of stage g + 2
// (24 VGPRs) at the top of stage g, writes them into the free stage buffer at the top of stage g + 1 (ds_write_b128: 13 cycles
// per KiB), and they are read as stage g + 2 -- two LDS buffers, registers as the third, one s_barrier per stage, hipcc's own
// vmcnt bookkeeping (the loads are a full stage old when their registers are read; a second register set, two stages ahead, was
// slower: 2.1 ms).
template <int WMW>
__global__ __launch_bounds__(128 * WMW, 2) void gemm_tall_bf16_kernel(const TgParams P) {
  using C = TgCfg<WMW>;
  constexpr int TG_BM = C::BM, TG_A = C::A, TG_STAGE = C::STAGE, NP = C::NP;
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
  // ---- pieces of this wave: piece p = wave + NW q (q = 0..NP-1): p < PA (q < 4) -> A rows 8p.., else W rows 8(p-PA)..; a lane
  // takes the 16-byte chunk (lane & 7) of row (lane >> 3) and stores it at slot chunk ^ ((row >> 1) & 7) of its 128-byte LDS row
  uint32_t voff[NP], loff[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int p = wave + C::NW * q;
    const int row = (q < 4 ? 8 * p : 8 * (p - C::PA)) + (lane >> 3), chunk = lane & 7;
    voff[q] = (uint32_t)(row * (q < 4 ? (int)P.lda : (int)P.ldw) * 2 + chunk * 16);
    loff[q] = (uint32_t)((q < 4 ? 0 : TG_A) + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
  }
  char* const tab = smem + 2 * TG_STAGE;

  float4 pr[NP];                                    // the pieces in flight
  float4 tr = make_float4(0.f, 0.f, 0.f, 0.f);      // wave 0: the row-group table piece in flight
  int ft = 0, fks = 0, fcount = 0;                  // fetch cursor: tile number (of mine), k-stage, stages fetched so far
  auto load_stage = [&]() {                         // past the last stage: zero-byte resources, the loads return zeros
    const bool on = fcount < nstages;
    const int tile = first + ft * (int)gridDim.x;
    const int mt = tile / P.tiles_n, nt = tile - mt * P.tiles_n;
    const char* a = P.A + (int64_t)mt * TG_BM * P.lda * 2;
    const int64_t left = P.A_end - a;                       // rows past M read zeros (range check)
    const __amdgpu_buffer_rsrc_t ra = buf_rsrc(a, !on || left <= 0 ? 0u : left > 0xffff0000LL ? 0xffff0000u : (uint32_t)left);
    const __amdgpu_buffer_rsrc_t rw = buf_rsrc(P.W + (int64_t)nt * TG_BN * P.ldw * 2, on ? 0xffff0000u : 0u);
    const uint32_t ko = (uint32_t)(fks * TG_BK * 2);
#pragma unroll
    for (int q = 0; q < NP; ++q) pr[q] = buf_load16(q < 4 ? ra : rw, voff[q], ko);
    ++fcount;
    if (++fks == P.nk) { fks = 0; ++ft; }
  };
  auto store_stage = [&](int buf) {
    char* const st = smem + buf * TG_STAGE;
#pragma unroll
    for (int q = 0; q < NP; ++q) *reinterpret_cast<float4*>(st + loff[q]) = pr[q];
  };
  // row-group table of a tile (wave 0): lanes 0-31 the tile's first group, 32-63 the next one, 4 columns each
  auto load_table = [&](int tile) {
    const int mt = tile / P.tiles_n, nt = tile - mt * P.tiles_n;
    const int g0 = (mt * TG_BM) / P.rg_div + (lane >> 5);
    const __amdgpu_buffer_rsrc_t rr = buf_rsrc(P.rg);
    tr = buf_load16(rr, g0 < P.rg_groups ? (uint32_t)(((int64_t)g0 * P.rg_ld + nt * TG_BN + 4 * (lane & 31)) * 4) : BUF_OOB);
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

  // prologue: stage 0 into buffer 0, stage 1 into the registers
  load_stage();
  store_stage(0);
  load_stage();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int buf = 0;
  for (int t = 0; t < my_tiles; ++t) {
    const int tile = first + t * (int)gridDim.x;
    const int mt = tile / P.tiles_n, nt = tile - mt * P.tiles_n;
    const int m0 = mt * TG_BM, n0 = nt * TG_BN;
    for (int ks = 0; ks < P.nk; ++ks) {
      __builtin_amdgcn_s_barrier();                 // stage (t, ks) is in LDS for everybody; the other buffer is free
      // the registers hold the next stage (loaded a stage ago): into the free buffer, then the loads of the stage after it
      // the registers hold the next stage (loaded a stage ago): into the free buffer, then the loads of the stage after it
      if (!TG_DBG(2)) store_stage(buf ^ 1);
      if (wave == 0 && P.rg) {
        // the tile's row-group table: loaded with stage nk - 3, stored with stage nk - 2, so that the barrier of the tile's last
        // stage shows it to everybody before the epilogue (its previous reader, the previous tile's epilogue, is long done)
        if (ks == P.nk - 2) *reinterpret_cast<float4*>(tab + lane * 16) = tr;
        if (ks == P.nk - 3) load_table(tile);
      }
      if (!TG_DBG(2)) load_stage();
      __builtin_amdgcn_sched_barrier(0);
      const char* const st = smem + buf * TG_STAGE;
      // fragments of k-step s + 1 are read while the MFMAs of k-step s run.  The accumulators are TRANSPOSED (W rows as the A
      // operand): lane (r, h) holds row r of the 32-row block, 4 consecutive columns per register group -- the epilogue packs
      // them to 8 bytes and never goes through LDS
      bf16x8 a[2][2], b[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[0][i] = *reinterpret_cast<const bf16x8*>(st + aoff[i][0]);
        b[0][i] = *reinterpret_cast<const bf16x8*>(st + boff[i][0]);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < 3) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            a[(s + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(st + aoff[i][s + 1]);
            b[(s + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(st + boff[i][s + 1]);
          }
        }
        if (!TG_DBG(4)) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[s & 1][j], a[s & 1][i], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(a[s & 1][i]), "v"(b[s & 1][i]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // the ds_writes above must have landed before the next barrier lets others read them
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      buf ^= 1;
    }
    if (TG_DBG(1)) {
      if (acc[0][0][0] == 123.456f) P.C[0] = 1;      // keeps the accumulators alive
      zero_acc();
      continue;
    }
    // ---- epilogue of the tile: (+|*) row-group term, ReLU, bf16, straight from the registers.  Lane (r, h) holds, of row r, the
    // columns 8 c + 4 h .. + 3 of every 8-column chunk c; one v_permlane32_swap per register hands the lower lanes the whole
    // even chunks and the upper lanes the whole odd ones: 16-byte stores, four per 32-row block (the LDS round trip -- 64
    // two-byte ds_writes per wave and tile -- was 0.5-0.9 ms of this kernel's 1.9)
    const bool mul = P.rg_op != 0, relu = P.relu != 0;
    const int g0 = P.rg ? m0 / P.rg_div : 0;
    const int boundary = (g0 + 1) * P.rg_div;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = m0 + wm * 64 + 32 * i + r;
      const char* const tb = tab + (row >= boundary ? 512 : 0) + (wn * 64 + 4 * h) * 4;
      uint32_t G[8][2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          float v[4];
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) v[e4] = acc[i][j][4 * q4 + e4];
          if (P.rg) {
            const float4 tt = *reinterpret_cast<const float4*>(tb + (32 * j + 8 * q4) * 4);
            if (mul) { v[0] *= tt.x; v[1] *= tt.y; v[2] *= tt.z; v[3] *= tt.w; }
            else { v[0] += tt.x; v[1] += tt.y; v[2] += tt.z; v[3] += tt.w; }
          }
          if (relu) {
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) v[e4] = fmaxf(v[e4], 0.f);
          }
          G[4 * j + q4][0] = pack_bf16x2(v[0], v[1]);
          G[4 * j + q4][1] = pack_bf16x2(v[2], v[3]);
        }
      const __amdgpu_buffer_rsrc_t ro = TG_DBG(8) ? buf_rsrc(P.C + (int64_t)(wm * 64 + 32 * i) * P.ldc + wn * 64)
                                                  : buf_rsrc(P.C + (int64_t)(m0 + wm * 64 + 32 * i) * P.ldc + n0 + wn * 64);
      const uint32_t vo = row < P.M ? (uint32_t)r * (uint32_t)P.ldc * 2u + 16u * h : BUF_OOB;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const auto s0 = __builtin_amdgcn_permlane32_swap(G[2 * k][0], G[2 * k + 1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(G[2 * k][1], G[2 * k + 1][1], false, false);
        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
        if (TG_DBG(16)) asm volatile("" ::"v"(o));
        else __builtin_amdgcn_raw_buffer_store_b128(o, ro, (int)vo, 32 * k, 0);
      }
    }
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
  P.tiles_n = N / TG_BN; P.nk = K / TG_BK;
  {
    const char* e = getenv("VQA_TALL_DBG");
    P.dbg = e ? atoi(e) : 0;
  }
  hipStream_t s = (hipStream_t)stream;
  set_launch_tag(tag);
  ProfScope prof(VQA_K_GEMM, s);
  // 256 x 128 tiles, one workgroup of eight waves per CU (1.68 ms at 1.49 M x 1024 x 256); VQA_TALL_BM=128: 128 x 128 tiles on two
  // workgroups of four waves per CU (1.82 ms: the two do not hide each other's phases, and W is fetched twice as often)
  static const int bm256 = [] { const char* e = getenv("VQA_TALL_BM"); return e && atoi(e) == 128 ? 0 : 1; }();
  if (bm256) {
    P.tiles_m = (M + 255) / 256;
    int rc = ensure_dyn_smem(reinterpret_cast<const void*>(gemm_tall_bf16_kernel<4>), TgCfg<4>::LDS, "attr(gemm_tall_bf16)");
    if (rc) return rc;
    const int tiles = P.tiles_m * P.tiles_n;
    hipLaunchKernelGGL(gemm_tall_bf16_kernel<4>, dim3(tiles < 256 ? tiles : 256), dim3(512), TgCfg<4>::LDS, s, P);
  } else {
    P.tiles_m = (M + 127) / 128;
    int rc = ensure_dyn_smem(reinterpret_cast<const void*>(gemm_tall_bf16_kernel<2>), TgCfg<2>::LDS, "attr(gemm_tall_bf16)");
    if (rc) return rc;
    const int tiles = P.tiles_m * P.tiles_n;
    hipLaunchKernelGGL(gemm_tall_bf16_kernel<2>, dim3(tiles < 512 ? tiles : 512), dim3(256), TgCfg<2>::LDS, s, P);
  }
  return check_hip(hipGetLastError(), "gemm_tall_bf16 launch");
}

}  // extern "C"
