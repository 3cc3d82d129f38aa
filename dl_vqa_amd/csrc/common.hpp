// Shared host/device helpers for the VQA HIP kernel library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vqa_hip.h"

namespace vqa {

// ---------------------------------------------------------------- errors
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

#define VQA_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      ::vqa::set_error(__VA_ARGS__);           \
      return VQA_ERR_INVALID;                  \
    }                                          \
  } while (0)

// ---------------------------------------------------------------- launch plumbing
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) for a kernel, once per (device, kernel); thread-safe.
int ensure_dyn_smem(const void* kernel, int bytes, const char* what);

// VQA_* environment knobs (diagnostics / forced tile variants for the parity tests).  Read ONCE, when the
// library is first used; vqa_reload_knobs() re-reads them (tests that switch variants inside one process).
//   -1 = unset (the planner decides)
struct Knobs {
  int split_target;       // VQA_SPLIT_TARGET: resident-workgroup target of the split-K planner (default 512)
  int big_tiles;          // VQA_BIG_TILES 0/1/2/3
  int persistent;         // VQA_PERSISTENT 0/1
  int weight_stationary;  // VQA_WEIGHT_STATIONARY 0/1
  int wgrad_192;          // VQA_WGRAD_192 0/1
  int wgrad_384;          // VQA_WGRAD_384 0/1
  int conv_chunk;         // VQA_CONV_CHUNK: images per conv launch (tests)
};
const Knobs& knobs();

// ---------------------------------------------------------------- profiling hook
// When a kernel id is armed (vqa_prof_arm), every launch of that kernel is bracketed by HIP
// events recorded on the launch stream; vqa_prof_read() returns count and total milliseconds.
void set_launch_tag(int tag);
struct ProfScope {
  int id;
  hipStream_t s;
  bool on;
  ProfScope(int id_, hipStream_t s_);
  ~ProfScope();
};

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
  // Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous range of logical
  // tile ids so neighbouring tiles (shared operand panels) meet in one L2.  Bijective for any nwg.
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// Counter-based dropout RNG: keep(seed, site, idx) is a pure function, so the backward pass
// regenerates the mask instead of storing it.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float drop_scale(uint64_t seed, uint64_t idx, float p, float inv_keep) {
  // returns 0 (dropped) or 1/(1-p) (kept)
  uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
  uint32_t h = mix32(lo ^ (uint32_t)seed);
  h = mix32(h + hi * 0x9E3779B9U + (uint32_t)(seed >> 32));
  const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
  return u >= p ? inv_keep : 0.0f;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace vqa
