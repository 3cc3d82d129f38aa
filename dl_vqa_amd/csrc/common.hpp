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
// One 32-bit hash serves a PAIR of elements (idx >> 1), 16 random bits each: keep iff the element's 16-bit field is at or
// above floor(p * 65536) -- the drop probability is p to within 2^-16.  The hash was the VALU bound of the attention-stage
// kernels in train mode (two multiply-xorshift rounds per element: att_score_fwd ran at 2 TB/s against 4 TB/s in eval
// mode); vector code takes four consecutive elements from drop_scale4 (two hashes).
__device__ __forceinline__ uint32_t drop_hash(uint64_t seed, uint64_t pair) {
  const uint32_t lo = (uint32_t)pair, hi = (uint32_t)(pair >> 32);
  uint32_t h = mix32(lo ^ (uint32_t)seed);
  return mix32(h + hi * 0x9E3779B9U + (uint32_t)(seed >> 32));
}
__device__ __forceinline__ float drop_scale(uint64_t seed, uint64_t idx, float p, float inv_keep) {
  // returns 0 (dropped) or 1/(1-p) (kept)
  const uint32_t h = drop_hash(seed, idx >> 1);
  const uint32_t u = (idx & 1) ? (h >> 16) : (h & 0xffffu);
  return u >= (uint32_t)(p * 65536.0f) ? inv_keep : 0.0f;
}
// scales of elements idx .. idx + 3, idx a multiple of 4
__device__ __forceinline__ float4 drop_scale4(uint64_t seed, uint64_t idx, float p, float inv_keep) {
  const uint32_t h0 = drop_hash(seed, idx >> 1), h1 = drop_hash(seed, (idx >> 1) + 1);
  const uint32_t thr = (uint32_t)(p * 65536.0f);
  return make_float4((h0 & 0xffffu) >= thr ? inv_keep : 0.f, (h0 >> 16) >= thr ? inv_keep : 0.f,
                     (h1 & 0xffffu) >= thr ? inv_keep : 0.f, (h1 >> 16) >= thr ? inv_keep : 0.f);
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
