// Generic fp32 MFMA GEMM (vqa_gemm) + library-wide host plumbing (errors, profiling hook).
#include <stdarg.h>

#include <mutex>
#include <set>
#include <utility>
#include <vector>

#include <type_traits>
#include "gemm_core.hpp"
#include <map>
#include <mutex>
#include <utility>
#include "gemm_epilogue.hpp"

// This file is compiled six times (dl_vqa_amd/build.py): VQA_GEMM_PART = 0 is the host plumbing + the C ABI, parts 1-5 hold
// the kernels (1: 64x64 tiles; 2, 3: 128x128 tiles with A stored [M][K] / [K][M]; 4, 5: the same for 256x128) -- the fused
// epilogue's straight-line variants make a single translation unit with all of them take seven minutes to compile.
#ifndef VQA_GEMM_PART
#define VQA_GEMM_PART 0
#endif

namespace vqa {

#if VQA_GEMM_PART == 0
// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return VQA_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return VQA_ERR_HIP;
}

// ------------------------------------------------------------------ launch plumbing
constexpr int kMaxScratchPerLane = 256;   // bytes; the shipped kernels use 0-200 (tests/test_abi_cpu.py checks the code objects)
static std::mutex g_attr_mu;
static std::set<std::pair<int, const void*>> g_attr_done;
int ensure_dyn_smem(const void* kernel, int bytes, const char* what) {
  int dev = 0;
  int rc = check_hip(hipGetDevice(&dev), "hipGetDevice");
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_attr_mu);
  if (g_attr_done.count({dev, kernel})) return VQA_OK;
  rc = check_hip(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), what);
  if (rc) return rc;
  // Private-segment guard (round-3 root cause of the round-2 hang): a workgroup-barrier kernel whose waves
  // need scratch deadlocked once ~2 000 of its waves (512 bytes per lane: ~64 MB of scratch) were dispatched -- 176
  // workgroups of 8 waves ran, 248 and more hung, whatever the problem shape; the same epilogue in a kernel with 100 bytes
  // per lane ran at any grid.  Every kernel of this library meets at workgroup barriers, so a kernel that comes out of the
  // compiler with a large private segment is refused here, before it can take a GPU down (VQA_ALLOW_SCRATCH=1: experiments).
  hipFuncAttributes attr;
  rc = check_hip(hipFuncGetAttributes(&attr, kernel), "hipFuncGetAttributes");
  if (rc) return rc;
  if (attr.localSizeBytes > kMaxScratchPerLane && !(getenv("VQA_ALLOW_SCRATCH") && atoi(getenv("VQA_ALLOW_SCRATCH")) == 1)) {
    set_error("%s: the kernel needs %zu bytes of scratch per lane (limit %d): refused -- workgroup-barrier kernels with a "
              "large private segment hang gfx950 beyond ~2000 resident waves (DESIGN.md 7(5))", what, (size_t)attr.localSizeBytes,
              kMaxScratchPerLane);
    return VQA_ERR_INVALID;
  }
  g_attr_done.insert({dev, kernel});
  return VQA_OK;
}

static std::mutex g_knob_mu;
static Knobs g_knobs;
static bool g_knobs_read = false;
static int env_int(const char* name) {
  const char* e = getenv(name);
  return (e && *e) ? atoi(e) : -1;
}
static void read_knobs_locked() {
  g_knobs.split_target = env_int("VQA_SPLIT_TARGET");
  g_knobs.big_tiles = env_int("VQA_BIG_TILES");
  g_knobs.persistent = env_int("VQA_PERSISTENT");
  g_knobs.weight_stationary = env_int("VQA_WEIGHT_STATIONARY");
  g_knobs.wgrad_192 = env_int("VQA_WGRAD_192");
  g_knobs.wgrad_384 = env_int("VQA_WGRAD_384");
  g_knobs.conv_chunk = env_int("VQA_CONV_CHUNK");
  g_knobs_read = true;
}
const Knobs& knobs() {
  std::lock_guard<std::mutex> lk(g_knob_mu);
  if (!g_knobs_read) read_knobs_locked();
  return g_knobs;
}

// ------------------------------------------------------------------ profiling hook
static std::mutex g_prof_mu;
static uint32_t g_prof_mask = 0;
static int g_prof_tag = -1;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_ev;
static std::vector<std::pair<int, int>> g_prof_key;   // (family, tag) of each recorded event pair
static thread_local int g_launch_tag = -1;
void set_launch_tag(int tag) { g_launch_tag = tag; }

ProfScope::ProfScope(int id_, hipStream_t s_) : id(id_), s(s_), on(false) {
  if (g_prof_mask == 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!((g_prof_mask >> id) & 1u) || (g_prof_tag >= 0 && g_prof_tag != g_launch_tag)) return;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;       // never inside a stream capture
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
  hipEventRecord(a, s);
  g_prof_ev.emplace_back(a, b);
  g_prof_key.emplace_back(id, g_launch_tag);
  on = true;
}
ProfScope::~ProfScope() {
  if (!on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  hipEventRecord(g_prof_ev.back().second, s);
}

#endif  // VQA_GEMM_PART == 0

// Persistent variant (no split-K, single Raw set): min(tiles, resident slots) workgroups walk the tiles.
template <class Cfg, class AL, class BL>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::MIN_WAVES) void gemm_persistent_kernel(
    typename AL::Params pa, typename BL::Params pb, EpiParams pe, int tiles_m, int tiles_n, int nk, int Ktot,
    int order) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  auto origin = [&](int t, int& m0, int& n0) {
    int mt, nt;
    if (order == 1) { nt = t / tiles_m; mt = t - nt * tiles_m; } else { mt = t / tiles_n; nt = t - mt * tiles_n; }
    m0 = mt * Cfg::BM; n0 = nt * Cfg::BN;
  };
  gemm_persistent<Cfg, AL, BL, false>(
      xcd_swizzle(blockIdx.x, gridDim.x), gridDim.x, tiles_m * tiles_n, nk, Ktot, smem,
      [&](int t, AL& al, BL& bl) {
        int m0, n0;
        origin(t, m0, n0);
        al.init(pa, m0, loader_tid<Cfg>(), 0);
        bl.init(pb, n0, loader_tid<Cfg>(), 0);
      },
      [&](int t, f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
        int m0, n0;
        origin(t, m0, n0);
        gemm_epilogue<Cfg>(pe, acc, m0, n0, wm, wn, lane);
      }
#ifdef VQA_DIAG
      , pe.bar_dbg
#endif
      );
}

template <class Cfg, class AL, class BL>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::MIN_WAVES) void gemm_kernel(typename AL::Params pa, typename BL::Params pb,
                                                   EpiParams pe, int tiles_m, int tiles_n, int nk,
                                                   int ks_per_split, int Ktot, int order, int splits) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const TileCoord tc = tile_coord(tiles_m, tiles_n, order, splits);
  const int m0 = tc.mt * Cfg::BM, n0 = tc.nt * Cfg::BN;
  const int split = tc.split;
  const int ks0 = split * ks_per_split;
  const int ks1 = min(nk, ks0 + ks_per_split);
  f32x16 acc[Cfg::TM][Cfg::TN];
  acc_zero<Cfg>(acc);
  if (!gemm_mainloop<Cfg, AL, BL>(
          [&](AL& al, BL& bl) {
            al.init(pa, m0, loader_tid<Cfg>(), ks0);
            bl.init(pb, n0, loader_tid<Cfg>(), ks0);
          },
          [](AL&, BL&) {}, acc, ks0, ks1, Ktot, smem))
    return;

  float* slab = pe.slab ? pe.slab + (int64_t)split * pe.M * pe.N : nullptr;
  if (slab) {
    store_acc_tiles<Cfg>(acc, slab, pe.N, pe.M, pe.N, m0, n0, wm, wn, lane);
    return;
  }
  gemm_epilogue<Cfg>(pe, acc, m0, n0, wm, wn, lane);
}

// ------------------------------------------------------------------ host side
#if VQA_GEMM_PART == 0
// Workgroups of `kernel` that are resident at once on the current device: min(planned, occupancy query) per CU x CUs.
// Cached per (device, kernel).  <= 0: the query failed (vqa_last_error says why).
int persistent_slots(const void* kernel, int threads, int smem_bytes, int planned_per_cu) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> cache;
  int dev = 0;
  if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return -1;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair(dev, kernel);
  const auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int per_cu = 0, cus = 0;
  if (check_hip(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, (size_t)smem_bytes),
                "hipOccupancyMaxActiveBlocksPerMultiprocessor(gemm_persistent)"))
    return -1;
  if (check_hip(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute(CUs)")) return -1;
  if (per_cu < 1) {
    set_error("gemm_persistent: the kernel does not fit a CU (occupancy query says %d workgroups)", per_cu);
    return -1;
  }
  const int slots = cus * (per_cu < planned_per_cu ? per_cu : planned_per_cu);
  cache[key] = slots;
  return slots;
}

GemmPlan plan_gemm(int M, int N, int K, int bk) {
  GemmPlan p;
  const int nk = (K + bk - 1) / bk;
  const int t128 = ((M + 127) / 128) * ((N + 127) / 128);
  const int t64 = ((M + 63) / 64) * ((N + 63) / 64);
  int splits = 1;
  // Two workgroups fit a CU: 512 resident slots.  A launch that fills them exactly keeps two MFMA waves on
  // every SIMD (one covers the other's barriers); 384 workgroups left a quarter of the slots empty
  // (v_conv dW 0.90 -> 0.72 ms with 24 -> 32 splits), more than 512 would run a second, nearly empty round.
  const Knobs& kn = knobs();
  const int target = kn.split_target > 0 ? kn.split_target : 512;
  if (t128 >= 200) {
    p.big = 1;
    // 200..511 big tiles with a long K: two splits double the resident workgroups (LSTM dW_hh: 256 tiles)
    if (t128 * 2 <= target && nk >= 64) splits = target / t128;
  } else if (t64 >= 200) p.big = 0;
  else {
    // few big tiles + many splits = short K loops dominated by prologue/epilogue (LSTM dh GEMM, M = 256:
    // 16 tiles x 24 splits of 6 K-steps ran at 21 % of peak): below 64 big tiles use 64x64 tiles, whose
    // 4x larger tile count needs 4x fewer splits
    // ... unless K is long enough that every split of the big tiling still runs >= 16 K-steps
    // (v_conv dW: K = B*P = 173k -> 16 tiles x 32 splits x 169 K-steps)
    const int big_splits = target / t128 > 1 ? target / t128 : 1;
    p.big = (M >= 128 && N >= 128 && (t128 >= 64 || nk / big_splits >= 16)) ? 1 : 0;
    const int tiles = p.big ? t128 : t64;
    const int max_splits = nk / 4 > 1 ? nk / 4 : 1;
    splits = target / tiles > 1 ? target / tiles : 1;
    if (splits > max_splits) splits = max_splits;
    if (splits > 64) splits = 64;
  }
  // 256x128 tiles (8 MFMA waves, as the conv forward) measured 4-8 % SLOWER on the tall plain GEMMs
  // (v_conv fwd 1.29 vs 1.24 ms, dgrad 0.94 vs 0.87 ms): opt-in only (VQA_BIG_TILES=2)
  if (p.big && splits == 1 && kn.big_tiles == 2) p.big = 2;
  const int bm = p.big == 2 ? 256 : (p.big ? 128 : 64), bn = p.big ? 128 : 64;
  p.tiles_m = (M + bm - 1) / bm;
  p.tiles_n = (N + bn - 1) / bn;
  p.nk = nk;
  p.ks_per_split = (nk + splits - 1) / splits;
  p.splits = (nk + p.ks_per_split - 1) / p.ks_per_split;
  // skinny GEMM with a big B operand: keep each XCD on its own column slice of B (tile_coord order 1)
  p.order = (p.tiles_m <= 8 && p.tiles_n >= 16 && kn.weight_stationary != 0) ? 1 : 0;
  return p;
}

#endif  // VQA_GEMM_PART == 0

int persistent_slots(const void* kernel, int threads, int smem_bytes, int planned_per_cu);

template <class Cfg, class AL, class BL>
int launch_gemm(const typename AL::Params& pa, const typename BL::Params& pb, const EpiParams& pe,
                       const GemmPlan& p, int K, hipStream_t s) {
  using SL = SmemLayout<Cfg, AL::kTypeR, BL::kTypeR>;
  if constexpr (Cfg::BM * Cfg::BN > 64 * 64) {
    // Persistent tiles pay when K is short (<= 16 K-steps: dispatch + prologue + epilogue are then a large
    // share of a tile's life: v_conv forward 1.13 -> 1.04 ms, LSTM input GEMM 0.116 -> 0.105 ms); with long K
    // the static tile striding loses more to imbalance than it saves (v_conv dgrad 0.82 -> 0.88 ms), so those
    // keep one workgroup per tile and the hardware's dynamic dispatch.  VQA_PERSISTENT=0/1 forces the choice.
    const int pt = knobs().persistent;
    const bool persistent = pt >= 0 ? pt == 1 : p.nk <= 16;
    if (p.splits == 1 && persistent) {
      auto pk = gemm_persistent_kernel<Cfg, AL, BL>;
      int rc = ensure_dyn_smem(reinterpret_cast<const void*>(pk), SL::BYTES, "hipFuncSetAttribute(gemm_persistent)");
      if (rc) return rc;
      // resident slots from the occupancy the runtime reports for THIS kernel (registers, scratch and LDS as compiled), not
      // from the LDS layout alone: a variant that needs more registers than planned then gets a smaller grid instead of a
      // second, queued round of workgroups.  (No workgroup ever waits for another one, so co-residency is a matter of
      // speed here, never of progress.)
      const int slots = persistent_slots(reinterpret_cast<const void*>(pk), Cfg::THREADS, SL::BYTES, SL::WG_PER_CU);
      if (slots <= 0) return VQA_ERR_HIP;
      const int tiles = p.tiles_m * p.tiles_n;
      hipLaunchKernelGGL(pk, dim3(tiles < slots ? tiles : slots), dim3(Cfg::THREADS), SL::BYTES, s, pa, pb, pe,
                         p.tiles_m, p.tiles_n, p.nk, K, p.order);
      return check_hip(hipGetLastError(), "gemm_persistent_kernel launch");
    }
  }
  auto kern = gemm_kernel<Cfg, AL, BL>;
  {
    int rc = ensure_dyn_smem(reinterpret_cast<const void*>(kern), SL::BYTES, "hipFuncSetAttribute(gemm)");
    if (rc) return rc;
  }
  dim3 grid(p.tiles_m * p.tiles_n * p.splits);
  hipLaunchKernelGGL(kern, grid, dim3(Cfg::THREADS), SL::BYTES, s, pa, pb, pe, p.tiles_m, p.tiles_n, p.nk,
                     p.ks_per_split, K, p.order, p.splits);
  return check_hip(hipGetLastError(), "gemm_kernel launch");
}

using Cfg128 = TileCfg<128, 128, 2, 2>;
using Cfg256 = TileCfg<256, 128, 4, 2>;
using Cfg64 = TileCfg<64, 64, 2, 2>;

template <class Cfg>
int dispatch_gemm(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB,
                  const EpiParams& pe, const GemmPlan& p, int M, int N, int K, hipStream_t s) {
  using AR = PlainR<Cfg::NVA, Cfg::LT>; using AC = PlainC<Cfg::NVA, Cfg::LT>;
  using BR = PlainR<Cfg::NVB, Cfg::LT>; using BC = PlainC<Cfg::NVB, Cfg::LT>;
  if (!transA && transB) return launch_gemm<Cfg, AR, BR>({A, lda, M, K}, {B, ldb, N, K}, pe, p, K, s);
  if (!transA && !transB) return launch_gemm<Cfg, AR, BC>({A, lda, M, K}, {B, ldb, N, K}, pe, p, K, s);
  if (transA && transB) return launch_gemm<Cfg, AC, BR>({A, lda, M, K}, {B, ldb, N, K}, pe, p, K, s);
  return launch_gemm<Cfg, AC, BC>({A, lda, M, K}, {B, ldb, N, K}, pe, p, K, s);
}

// the kernels of launch_gemm<Cfg, A loader, B loader> live in the part that owns the combination
#define VQA_GEMM_LAUNCH(KW, CFG, AL, BL)                                                                           \
  KW template int launch_gemm<CFG, AL<CFG::NVA, CFG::LT>, BL<CFG::NVB, CFG::LT>>(                                   \
      const typename AL<CFG::NVA, CFG::LT>::Params&, const typename BL<CFG::NVB, CFG::LT>::Params&, const EpiParams&, \
      const GemmPlan&, int, hipStream_t);
#define VQA_GEMM_OWN(PART, CFG, AL, BL) VQA_GEMM_LAUNCH(, CFG, AL, BL)
#define VQA_GEMM_EXT(PART, CFG, AL, BL) VQA_GEMM_LAUNCH(extern, CFG, AL, BL)
#define VQA_GEMM_COMBOS(X1, X2, X3, X4, X5)                                                                        \
  X1(1, Cfg64, PlainR, PlainR) X1(1, Cfg64, PlainR, PlainC) X1(1, Cfg64, PlainC, PlainR) X1(1, Cfg64, PlainC, PlainC)  \
  X2(2, Cfg128, PlainR, PlainR) X2(2, Cfg128, PlainR, PlainC) X3(3, Cfg128, PlainC, PlainR) X3(3, Cfg128, PlainC, PlainC) \
  X4(4, Cfg256, PlainR, PlainR) X4(4, Cfg256, PlainR, PlainC) X5(5, Cfg256, PlainC, PlainR) X5(5, Cfg256, PlainC, PlainC)
#if VQA_GEMM_PART == 1
VQA_GEMM_COMBOS(VQA_GEMM_OWN, VQA_GEMM_EXT, VQA_GEMM_EXT, VQA_GEMM_EXT, VQA_GEMM_EXT)
#elif VQA_GEMM_PART == 2
VQA_GEMM_COMBOS(VQA_GEMM_EXT, VQA_GEMM_OWN, VQA_GEMM_EXT, VQA_GEMM_EXT, VQA_GEMM_EXT)
#elif VQA_GEMM_PART == 3
VQA_GEMM_COMBOS(VQA_GEMM_EXT, VQA_GEMM_EXT, VQA_GEMM_OWN, VQA_GEMM_EXT, VQA_GEMM_EXT)
#elif VQA_GEMM_PART == 4
VQA_GEMM_COMBOS(VQA_GEMM_EXT, VQA_GEMM_EXT, VQA_GEMM_EXT, VQA_GEMM_OWN, VQA_GEMM_EXT)
#elif VQA_GEMM_PART == 5
VQA_GEMM_COMBOS(VQA_GEMM_EXT, VQA_GEMM_EXT, VQA_GEMM_EXT, VQA_GEMM_EXT, VQA_GEMM_OWN)
#else
VQA_GEMM_COMBOS(VQA_GEMM_EXT, VQA_GEMM_EXT, VQA_GEMM_EXT, VQA_GEMM_EXT, VQA_GEMM_EXT)
#endif

}  // namespace vqa

#if VQA_GEMM_PART == 0
using namespace vqa;
#ifdef VQA_DIAG
static unsigned long long* g_bar_dbg = nullptr;
#endif

extern "C" {

int vqa_abi_version(void) { return VQA_ABI_VERSION; }

#ifdef VQA_DIAG
/* diagnostic build only (tools/diag_barriers.py): device buffer of 4 x uint64 that the persistent GEMM kernels add their
 * per-role barrier counts to ([0] loader-wave barriers, [1] loader waves, [2] MFMA-wave barriers, [3] MFMA waves) */
int vqa_diag_barrier_buffer(void* dev_ptr) { g_bar_dbg = static_cast<unsigned long long*>(dev_ptr); return 0; }
#endif
const char* vqa_last_error(void) { return g_err; }

int vqa_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

int vqa_reload_knobs(void) {
  std::lock_guard<std::mutex> lk(g_knob_mu);
  read_knobs_locked();
  return VQA_OK;
}

int vqa_prof_arm_mask(uint32_t mask, int tag) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& e : g_prof_ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  g_prof_ev.clear();
  g_prof_key.clear();
  g_prof_mask = mask;
  g_prof_tag = tag;
  return VQA_OK;
}

int vqa_prof_arm(int kernel_id, int tag) {
  const uint32_t all = (1u << VQA_K_COUNT) - 1u;
  return vqa_prof_arm_mask(kernel_id < 0 ? 0u : (kernel_id >= VQA_K_COUNT ? all : (1u << kernel_id)), tag);
}

int vqa_prof_read_groups(int* ids, int* tags, int* launches, float* total_ms, int cap) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int n = 0;
  for (size_t e = 0; e < g_prof_ev.size(); ++e) {
    if (hipEventSynchronize(g_prof_ev[e].second) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_prof_ev[e].first, g_prof_ev[e].second) != hipSuccess) continue;
    int g = 0;
    while (g < n && g < cap && !(ids[g] == g_prof_key[e].first && tags[g] == g_prof_key[e].second)) ++g;
    if (g >= cap) continue;
    if (g == n) { ids[g] = g_prof_key[e].first; tags[g] = g_prof_key[e].second; launches[g] = 0; total_ms[g] = 0.f; ++n; }
    launches[g] += 1;
    total_ms[g] += ms;
  }
  return n;
}

int vqa_prof_read(int* launches, float* total_ms) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  float tot = 0.f;
  int n = 0;
  for (auto& e : g_prof_ev) {
    if (hipEventSynchronize(e.second) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { tot += ms; ++n; }
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = tot;
  return VQA_OK;
}

int64_t vqa_gemm_workspace_bytes(int M, int N, int K) {
  const GemmPlan p = plan_gemm(M, N, K);
  return p.splits > 1 ? (int64_t)p.splits * M * N * 4 : 0;
}

int vqa_gemm(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C,
             int64_t ldc, int M, int N, int K, const float* bias1, const float* bias2,
             const float* rowgroup, int64_t rg_ld, int rg_div, int rg_op, int relu, int accumulate,
             float* aux, float* workspace, int64_t workspace_bytes, int tag, vqa_stream_t stream) {
  VQA_REQUIRE(A && B && C, "vqa_gemm: null operand");
  VQA_REQUIRE(M > 0 && N > 0 && K > 0, "vqa_gemm: bad shape M=%d N=%d K=%d", M, N, K);
  VQA_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && lda % 4 == 0 && ldb % 4 == 0,
              "vqa_gemm: A/B must be 16-byte aligned with leading dimensions multiple of 4 (lda=%lld ldb=%lld)",
              (long long)lda, (long long)ldb);
  VQA_REQUIRE(lda < (1 << 21) && ldb < (1 << 21) && ldc < (1 << 21),
              "vqa_gemm: leading dimensions must be below 2^21 (lda=%lld ldb=%lld ldc=%lld)", (long long)lda,
              (long long)ldb, (long long)ldc);
  VQA_REQUIRE(!rowgroup || rg_div > 0, "vqa_gemm: rg_div must be positive");
  hipStream_t s = (hipStream_t)stream;
  const GemmPlan p = plan_gemm(M, N, K);
  EpiParams pe{C, ldc, M, N, bias1, bias2, rowgroup, rg_ld, rg_div, rg_op, relu, accumulate, aux, nullptr, nullptr};
#ifdef VQA_DIAG
  pe.bar_dbg = g_bar_dbg;
#endif
#ifdef VQA_EXP_EPI_DROPOUT
  {   // reconstruction of the withdrawn experiment: every row-group GEMM with ReLU drops 30 % in its epilogue
    const bool on = rowgroup != nullptr && relu;
    pe.drop_p = on ? 0.3f : 0.f; pe.drop_inv = 1.0f / 0.7f; pe.drop_seed = 0x1234567887654321ull;
  }
#endif
  if (p.splits > 1) {
    const int64_t need = (int64_t)p.splits * M * N * 4;
    if (!workspace || workspace_bytes < need) {
      set_error("vqa_gemm: workspace %lld bytes < %lld needed", (long long)workspace_bytes, (long long)need);
      return VQA_ERR_WORKSPACE;
    }
    pe.slab = workspace;
  }
  set_launch_tag(tag);
  int rc;
  {
    ProfScope prof(VQA_K_GEMM, s);
    rc = p.big == 2 ? dispatch_gemm<Cfg256>(A, lda, transA, B, ldb, transB, pe, p, M, N, K, s)
         : p.big  ? dispatch_gemm<Cfg128>(A, lda, transA, B, ldb, transB, pe, p, M, N, K, s)
                  : dispatch_gemm<Cfg64>(A, lda, transA, B, ldb, transB, pe, p, M, N, K, s);
    if (rc) return rc;
    if (p.splits > 1) rc = launch_splitk_reduce(pe, p.splits, s);
  }
  return rc;
}

}  // extern "C"
#endif  // VQA_GEMM_PART == 0
