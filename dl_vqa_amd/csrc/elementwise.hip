// Memory-bound stages of the VQA hot path: dropout, L2-norm, embedding+tanh, LSTM cell, attention
// score / softmax / weighted sum, soft-target cross entropy, reductions, Adam.
// All are HBM-bound streaming kernels: 16-byte per-lane accesses, wave64 shuffles for reductions.
#include <hip/hip_fp16.h>
#include "common.hpp"

namespace vqa {

__device__ __forceinline__ uint32_t f2bf16_pair(float lo, float hi) {      // round to nearest even (v_cvt_pk_bf16_f32)
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

// 4 consecutive elements of an fp32 or bf16 row, as floats
template <bool XB>
__device__ __forceinline__ float4 ld4(const void* row, int c) {
  if (XB) {
    const uint2 w = reinterpret_cast<const uint2*>(row)[c];
    return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                       __uint_as_float(w.y & 0xffff0000u));
  }
  return reinterpret_cast<const float4*>(row)[c];
}
template <bool XB>
__device__ __forceinline__ void st4(void* row, int c, float4 v) {
  if (XB) reinterpret_cast<uint2*>(row)[c] = make_uint2(f2bf16_pair(v.x, v.y), f2bf16_pair(v.z, v.w));
  else reinterpret_cast<float4*>(row)[c] = v;
}

static inline int grid_for(int64_t n, int per_block, int cap = 8192) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// ------------------------------------------------------------------ dropout
__global__ void dropout_kernel(const float* x, float* y, int64_t n, float p, float inv_keep, uint64_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = x[i] * drop_scale(seed, (uint64_t)i, p, inv_keep);
}

// y += dropout(x): 4 elements per thread and iteration
__global__ void dropout_add_kernel(const float* x, float* y, int64_t n, float p, float inv_keep, uint64_t seed) {
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(x)[i];
    float4 b = reinterpret_cast<float4*>(y)[i];
    const float4 ds_ = drop_scale4(seed, (uint64_t)(4 * i), p, inv_keep);
    b.x += a.x * ds_.x; b.y += a.y * ds_.y; b.z += a.z * ds_.z; b.w += a.w * ds_.w;
    reinterpret_cast<float4*>(y)[i] = b;
  }
  if (blockIdx.x == 0 && threadIdx.x < n - 4 * n4) {
    const int64_t i = 4 * n4 + threadIdx.x;
    y[i] += x[i] * drop_scale(seed, (uint64_t)i, p, inv_keep);
  }
}

// ------------------------------------------------------------------ L2 norm (+ image dropout)
// one wave per row of C floats
// vdrop (optional, fp32 or bf16 by VB): dropout_{p2, seed2}(vn) written in the same pass
template <bool VB>
__global__ void l2norm_fwd_kernel(const float* pooled, float* vn, float* norm, int64_t rows, int C, float p,
                                  float inv_keep, uint64_t seed, void* vdrop, float p2, float inv_keep2, uint64_t seed2) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int nch = C >> 2;
  if (nch <= 64) {
    // C <= 256 (every config of the reference): a lane keeps its 4 channels in registers -- one pass over memory instead of
    // two -- and a wave has FOUR rows in flight (one 1-KB row per wave left the CU with 32 KB in flight: 3.9 TB/s).  Same
    // operations on every element as the general form below.
    constexpr int R = 4;
    const int c = lane;
    const bool cok = c < nch;
    for (int64_t r0 = R * wave; r0 < rows; r0 += R * nwaves) {
      float4 u[R];
#pragma unroll
      for (int k = 0; k < R; ++k)
        u[k] = (cok && r0 + k < rows) ? reinterpret_cast<const float4*>(pooled + (r0 + k) * C)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      float ss[R];
#pragma unroll
      for (int k = 0; k < R; ++k) {
        if (p > 0.f) {
          const uint64_t e = (uint64_t)(r0 + k) * C + 4 * c;
          const float4 ds_ = drop_scale4(seed, e, p, inv_keep);
          u[k].x *= ds_.x; u[k].y *= ds_.y; u[k].z *= ds_.z; u[k].w *= ds_.w;
        }
        ss[k] = 0.f;
        ss[k] += u[k].x * u[k].x + u[k].y * u[k].y + u[k].z * u[k].z + u[k].w * u[k].w;
      }
#pragma unroll
      for (int k = 0; k < R; ++k) ss[k] = wave_sum(ss[k]);
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const int64_t r = r0 + k;
        if (r >= rows) break;
        const float nrm = sqrtf(ss[k]);
        const float inv = 1.0f / (nrm + 1e-12f);
        if (lane == 0) norm[r] = nrm;
        if (!cok) continue;
        const float4 o = make_float4(u[k].x * inv, u[k].y * inv, u[k].z * inv, u[k].w * inv);
        reinterpret_cast<float4*>(vn + r * C)[c] = o;
        if (vdrop) {
          const uint64_t e = (uint64_t)r * C + 4 * c;
          float4 d = o;
          if (p2 > 0.f) {
            const float4 ds_ = drop_scale4(seed2, e, p2, inv_keep2);
            d.x *= ds_.x; d.y *= ds_.y; d.z *= ds_.z; d.w *= ds_.w;
          }
          if (VB) {
            uint2 w;
            w.x = f2bf16_pair(d.x, d.y); w.y = f2bf16_pair(d.z, d.w);
            reinterpret_cast<uint2*>(static_cast<uint16_t*>(vdrop) + r * C)[c] = w;
          } else {
            reinterpret_cast<float4*>(static_cast<float*>(vdrop) + r * C)[c] = d;
          }
        }
      }
    }
    return;
  }
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float4* src = reinterpret_cast<const float4*>(pooled + r * C);
    float ss = 0.f;
    for (int c = lane; c < nch; c += 64) {
      float4 u = src[c];
      if (p > 0.f) {
        const uint64_t e = (uint64_t)r * C + 4 * c;
        { const float4 ds_ = drop_scale4(seed, e, p, inv_keep); u.x *= ds_.x; u.y *= ds_.y; u.z *= ds_.z; u.w *= ds_.w; }
      }
      ss += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
    }
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    const float inv = 1.0f / (nrm + 1e-12f);
    if (lane == 0) norm[r] = nrm;
    float4* dst = reinterpret_cast<float4*>(vn + r * C);
    for (int c = lane; c < nch; c += 64) {
      float4 u = src[c];
      if (p > 0.f) {
        const uint64_t e = (uint64_t)r * C + 4 * c;
        { const float4 ds_ = drop_scale4(seed, e, p, inv_keep); u.x *= ds_.x; u.y *= ds_.y; u.z *= ds_.z; u.w *= ds_.w; }
      }
      const float4 o = make_float4(u.x * inv, u.y * inv, u.z * inv, u.w * inv);
      dst[c] = o;
      if (vdrop) {
        const uint64_t e = (uint64_t)r * C + 4 * c;
        float4 d = o;
        if (p2 > 0.f) {
          { const float4 ds_ = drop_scale4(seed2, e, p2, inv_keep2); d.x *= ds_.x; d.y *= ds_.y; d.z *= ds_.z; d.w *= ds_.w; }
        }
        if (VB) {
          uint2 w;
          w.x = f2bf16_pair(d.x, d.y); w.y = f2bf16_pair(d.z, d.w);
          reinterpret_cast<uint2*>(static_cast<uint16_t*>(vdrop) + r * C)[c] = w;
        } else {
          reinterpret_cast<float4*>(static_cast<float*>(vdrop) + r * C)[c] = d;
        }
      }
    }
  }
}

// OB 1 / 2: the gradient is stored as bf16 (bf16 path: it is the pooled gradient of the last conv block, staged as bf16 anyway);
// 2: channel-blocked [image][C/16][position][16] -- what the routed patches of the patch convolutions read (a lane's 4
// channels are 8 bytes of one block; the four rows of a workgroup are consecutive positions = 128 contiguous bytes per block).
// G > 0: the incoming gradient is JOINED here instead of being read -- d loss / d vn = the weighted-sum branch
// sum_g probs[b][g][p] * dout[b][g][:] (what att_apply_bwd wrote as dvn) + dropout-mask * dv_in (what dropout_add added):
// two full passes over a [B*P][C] fp32 tensor less per step.  Same operations in the same order as the separate kernels.
template <int OB, int G>
__global__ void l2norm_bwd_kernel(const float* dvn, const float* vn, const float* norm, void* dpooled_,
                                  int64_t rows, int C, float p, float inv_keep, uint64_t seed, int positions,
                                  const float* dout, int64_t dout_ld, const float* probs, const float* dv_in, float p_v,
                                  float inv_keep_v, uint64_t seed_v) {
  float* const dpooled = static_cast<float*>(dpooled_);
  uint16_t* const dpooled16 = static_cast<uint16_t*>(dpooled_);
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int nch = C >> 2;
  if (nch <= 64) {
    // C <= 256 (every config of the reference): a lane holds its 4 channels of the row in registers -- one pass over memory
    // instead of two -- and a wave has TWO rows in flight
    const int c = lane;
    const bool cok = c < nch;
    for (int64_t r0 = 2 * wave; r0 < rows; r0 += 2 * nwaves) {
      float4 a[2], b[2];
      int64_t img[2], pos[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t r = r0 + u < rows ? r0 + u : rows - 1;
        img[u] = G > 0 || OB == 2 ? r / positions : 0;
        pos[u] = r - img[u] * positions;
        b[u] = cok ? reinterpret_cast<const float4*>(vn + r * C)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (G == 0) {
          a[u] = cok ? reinterpret_cast<const float4*>(dvn + r * C)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
          float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
          float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
          if (cok) {
#pragma unroll
            for (int q = 0; q < (G > 0 ? G : 1); ++q) {
              const float pr = probs[(img[u] * G + q) * positions + pos[u]];
              const float4 go = reinterpret_cast<const float4*>(dout + img[u] * dout_ld + (int64_t)q * C)[c];
              o.x += pr * go.x; o.y += pr * go.y; o.z += pr * go.z; o.w += pr * go.w;
            }
            x = reinterpret_cast<const float4*>(dv_in + r * C)[c];
            if (p_v > 0.f) {
              const float4 ds_ = drop_scale4(seed_v, (uint64_t)r * C + 4 * c, p_v, inv_keep_v);
              x.x *= ds_.x; x.y *= ds_.y; x.z *= ds_.z; x.w *= ds_.w;
            }
          }
          a[u] = make_float4(o.x + x.x, o.y + x.y, o.z + x.z, o.w + x.w);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t r = r0 + u;
        float dot = a[u].x * b[u].x + a[u].y * b[u].y + a[u].z * b[u].z + a[u].w * b[u].w;
        dot = wave_sum(dot);
        if (r >= rows) break;
        const float n = norm[r];
        const float inv = 1.0f / (n + 1e-12f);
        const float k = n > 0.f ? dot * (n + 1e-12f) / n : 0.f;
        if (!cok) continue;
        float4 d = make_float4((a[u].x - b[u].x * k) * inv, (a[u].y - b[u].y * k) * inv, (a[u].z - b[u].z * k) * inv,
                               (a[u].w - b[u].w * k) * inv);
        if (p > 0.f) {
          const float4 ds_ = drop_scale4(seed, (uint64_t)r * C + 4 * c, p, inv_keep);
          d.x *= ds_.x; d.y *= ds_.y; d.z *= ds_.z; d.w *= ds_.w;
        }
        if (OB) {
          typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
          const bf2 lo = {(__bf16)d.x, (__bf16)d.y}, hi = {(__bf16)d.z, (__bf16)d.w};
          const uint2 o = make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
          if (OB == 2) *reinterpret_cast<uint2*>(dpooled16 + ((img[u] * (C >> 4) + (c >> 2)) * positions + pos[u]) * 16 + (c & 3) * 4) = o;
          else reinterpret_cast<uint2*>(dpooled16 + r * C)[c] = o;
        } else {
          reinterpret_cast<float4*>(dpooled + r * C)[c] = d;
        }
      }
    }
    return;
  }
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float4* g = reinterpret_cast<const float4*>(dvn + r * C);
    const float4* v = reinterpret_cast<const float4*>(vn + r * C);
    const int64_t img = G > 0 || OB == 2 ? r / positions : 0;
    const int64_t pos = r - img * positions;
    float pr[G > 0 ? G : 1];
    if (G > 0) {
#pragma unroll
      for (int q = 0; q < G; ++q) pr[q] = probs[(img * G + q) * positions + pos];
    }
    auto grad = [&](int c) -> float4 {
      if (G == 0) return g[c];
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < (G > 0 ? G : 1); ++q) {
        const float4 go = reinterpret_cast<const float4*>(dout + img * dout_ld + (int64_t)q * C)[c];
        o.x += pr[q] * go.x; o.y += pr[q] * go.y; o.z += pr[q] * go.z; o.w += pr[q] * go.w;
      }
      float4 x = reinterpret_cast<const float4*>(dv_in + r * C)[c];
      if (p_v > 0.f) {
        const float4 ds_ = drop_scale4(seed_v, (uint64_t)r * C + 4 * c, p_v, inv_keep_v);
        x.x *= ds_.x; x.y *= ds_.y; x.z *= ds_.z; x.w *= ds_.w;
      }
      return make_float4(o.x + x.x, o.y + x.y, o.z + x.z, o.w + x.w);
    };
    float dot = 0.f;
    for (int c = lane; c < nch; c += 64) {
      const float4 a = grad(c), b = v[c];
      dot += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    dot = wave_sum(dot);
    const float n = norm[r];
    const float inv = 1.0f / (n + 1e-12f);
    const float k = n > 0.f ? dot * (n + 1e-12f) / n : 0.f;
    float4* dst = reinterpret_cast<float4*>(dpooled + r * C);
    uint2* dst16 = reinterpret_cast<uint2*>(dpooled16 + r * C);
    for (int c = lane; c < nch; c += 64) {
      const float4 a = grad(c), b = v[c];
      float4 d = make_float4((a.x - b.x * k) * inv, (a.y - b.y * k) * inv, (a.z - b.z * k) * inv, (a.w - b.w * k) * inv);
      if (p > 0.f) {
        const uint64_t e = (uint64_t)r * C + 4 * c;
        { const float4 ds_ = drop_scale4(seed, e, p, inv_keep); d.x *= ds_.x; d.y *= ds_.y; d.z *= ds_.z; d.w *= ds_.w; }
      }
      if (OB) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        const bf2 lo = {(__bf16)d.x, (__bf16)d.y}, hi = {(__bf16)d.z, (__bf16)d.w};
        const uint2 o = make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
        if (OB == 2) {
          *reinterpret_cast<uint2*>(dpooled16 + ((img * (C >> 4) + (c >> 2)) * positions + pos) * 16 + (c & 3) * 4) = o;
        } else {
          dst16[c] = o;
        }
      } else {
        dst[c] = d;
      }
    }
  }
}

// ------------------------------------------------------------------ embedding + dropout + tanh
// Token ids outside [0, V) (nn.Embedding raises for them, models/model.py:155) are COUNTED in *bad (device
// int32, optional) and treated as a zero embedding row in forward and backward alike; the host mirror raises
// from the counter (dl_vqa_amd/model.py).
__global__ void embed_tanh_fwd_kernel(const int64_t* q, const float* emb, float* x, int B, int T, int E, int V,
                                      float p, float inv_keep, uint64_t seed, int* bad) {
  const int64_t total = (int64_t)T * B * E;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % E);
    const int64_t tb = i / E;
    const int b = (int)(tb % B), t = (int)(tb / B);
    const int64_t tok = q[(int64_t)b * T + t];
    const bool ok = tok >= 0 && tok < V;
    if (!ok && e == 0 && bad) atomicAdd(bad, 1);
    float v = ok ? emb[tok * E + e] : 0.f;
    if (p > 0.f) v *= drop_scale(seed, ((uint64_t)b * T + t) * E + e, p, inv_keep);
    x[i] = tanhf(v);
  }
}

// One workgroup per vocabulary row v: scans the B*T token slots in index order, collects the slots whose token
// is v (ballot-ordered compaction through LDS) and adds their gradient rows in that fixed order, so the result
// is bitwise reproducible (no float atomics); rows that no slot references -- and row 0, the padding index,
// which receives no gradient -- are written as zeros, so demb needs no memset.
__global__ __launch_bounds__(256) void embed_tanh_bwd_kernel(const int64_t* q, const float* x, const float* dx,
                                                             float* demb, int B, int T, int E, int V, float p,
                                                             float inv_keep, uint64_t seed) {
  __shared__ int hits[256];
  __shared__ int wcount[4];
  const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = B * T;
  for (int e0 = 0; e0 < E; e0 += 1024) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (v != 0) {
      for (int s0 = 0; s0 < N; s0 += 256) {
        const int s = s0 + tid;
        const bool hit = s < N && q[s] == (int64_t)v;
        const int total = __syncthreads_count(hit);       // uniform; also fences the previous round's hits[]
        if (total == 0) continue;
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wcount[w];
        if (hit) hits[base + __popcll(bal & ((1ull << lane) - 1ull))] = s;
        __syncthreads();
        for (int h = 0; h < total; ++h) {
          const int sl = hits[h];
          const int b = sl / T, t = sl - b * T;
          const int64_t row = ((int64_t)t * B + b) * E;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int e = e0 + tid + 256 * k;
            if (e < E) {
              const float xv = x[row + e];
              float g = dx[row + e] * (1.f - xv * xv);
              if (p > 0.f) g *= drop_scale(seed, ((uint64_t)b * T + t) * E + e, p, inv_keep);
              acc[k] += g;
            }
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = e0 + tid + 256 * k;
      if (e < E) demb[(int64_t)v * E + e] = acc[k];
    }
  }
}

// The same result from a slot INDEX: the kernel above scans all B*T slots once per vocabulary row -- O(V * B*T)
// whatever the number of rows that are hit (hundreds of millions of reads at a real vocabulary with the stress shape).  With a
// workspace the slots are first binned by token (integer atomics: the bin CONTENTS are deterministic, their order is not), and a
// row's workgroup sorts its own bin ascending before it sums, so the summation order -- ascending slot index -- and therefore
// every bit of demb equal the scanning kernel's.  Bins above 1 024 slots (a token in more than 1 024 slots of one batch) fall
// back to the scan for that row only.
__global__ void embed_hist_kernel(const int64_t* q, int N, int V, int* counts) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N) return;
  const int64_t tok = q[s];
  if (tok > 0 && tok < V) atomicAdd(&counts[tok], 1);          // row 0 = padding index: no gradient
}
// offsets[v] = sum of counts[0..v), offsets[V] = total; cursor = a second copy for the fill pass.  One block of 1024 threads.
__global__ __launch_bounds__(1024) void embed_scan_kernel(const int* counts, int* offsets, int* cursor, int V) {
  __shared__ int part[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int v0 = 0; v0 < V; v0 += 1024) {
    const int v = v0 + threadIdx.x;
    const int cnt = v < V ? counts[v] : 0;
    part[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                        // inclusive Hillis-Steele scan of the chunk
      const int add = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    const int excl = carry + part[threadIdx.x] - cnt;
    if (v < V) { offsets[v] = excl; cursor[v] = excl; }
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) offsets[V] = carry;
}
__global__ void embed_fill_kernel(const int64_t* q, int N, int V, int* cursor, int* slots) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N) return;
  const int64_t tok = q[s];
  if (tok > 0 && tok < V) slots[atomicAdd(&cursor[tok], 1)] = s;
}
constexpr int EMB_BIN_CAP = 1024;
__global__ __launch_bounds__(256) void embed_tanh_bwd_binned_kernel(const int64_t* q, const float* x, const float* dx, float* demb,
                                                                    int B, int T, int E, int V, float p, float inv_keep,
                                                                    uint64_t seed, const int* offsets, const int* slots) {
  __shared__ int bin[EMB_BIN_CAP];
  __shared__ int hits[256];
  __shared__ int wcount[4];
  const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = B * T;
  const int o0 = offsets[v], n = offsets[v + 1] - o0;
  if (n == 0) {                                                  // rows nothing references (and row 0): zeros
    for (int e = tid; e < E; e += 256) demb[(int64_t)v * E + e] = 0.f;
    return;
  }
  const bool binned = n <= EMB_BIN_CAP;
  if (binned) {
    int np2 = 2;
    while (np2 < n) np2 <<= 1;
    for (int i = tid; i < np2; i += 256) bin[i] = i < n ? slots[o0 + i] : 0x7fffffff;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)                           // bitonic sort, ascending
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < np2; i += 256) {
          const int l = i ^ j;
          if (l > i) {
            const int a = bin[i], bb = bin[l];
            const bool up = (i & k) == 0;
            if ((a > bb) == up) { bin[i] = bb; bin[l] = a; }
          }
        }
        __syncthreads();
      }
  }
  for (int e0 = 0; e0 < E; e0 += 1024) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    auto add_slot = [&](int sl) {
      const int b = sl / T, t = sl - b * T;
      const int64_t row = ((int64_t)t * B + b) * E;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = e0 + tid + 256 * k;
        if (e < E) {
          const float xv = x[row + e];
          float g = dx[row + e] * (1.f - xv * xv);
          if (p > 0.f) g *= drop_scale(seed, ((uint64_t)b * T + t) * E + e, p, inv_keep);
          acc[k] += g;
        }
      }
    };
    if (binned) {
      for (int h = 0; h < n; ++h) add_slot(bin[h]);
    } else {                                                     // an over-full bin: this row scans, as the kernel above
      for (int s0 = 0; s0 < N; s0 += 256) {
        const int s = s0 + tid;
        const bool hit = s < N && q[s] == (int64_t)v;
        const int total = __syncthreads_count(hit);
        if (total == 0) continue;
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wcount[w];
        if (hit) hits[base + __popcll(bal & ((1ull << lane) - 1ull))] = s;
        __syncthreads();
        for (int h = 0; h < total; ++h) add_slot(hits[h]);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = e0 + tid + 256 * k;
      if (e < E) demb[(int64_t)v * E + e] = acc[k];
    }
  }
}

// ------------------------------------------------------------------ LSTM cell
__global__ void lstm_cell_fwd_kernel(const float* xg, const float* hg, const float* c_in, const float* h_in,
                                     const int64_t* q_len, int t, float* gates, float* c_out, float* h_out,
                                     float* c_final, int64_t cf_ld, int B, int H) {
  const int64_t total = (int64_t)B * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / H), j = (int)(i - (int64_t)b * H);
    const int64_t g0 = (int64_t)b * 4 * H + j;
    const float cp = c_in[i], hp = h_in[i];
    float cn = cp, hn = hp;
    float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f;
    if ((int64_t)t < q_len[b]) {
      gi = sigmoidf_(xg[g0] + hg[g0]);
      gf = sigmoidf_(xg[g0 + H] + hg[g0 + H]);
      gg = tanhf(xg[g0 + 2 * H] + hg[g0 + 2 * H]);
      go = sigmoidf_(xg[g0 + 3 * H] + hg[g0 + 3 * H]);
      cn = gf * cp + gi * gg;
      hn = go * tanhf(cn);
    }
    gates[g0] = gi; gates[g0 + H] = gf; gates[g0 + 2 * H] = gg; gates[g0 + 3 * H] = go;
    c_out[i] = cn; h_out[i] = hn;
    if (c_final) c_final[(int64_t)b * cf_ld + j] = cn;
  }
}

__global__ void lstm_cell_bwd_kernel(const float* gates, const float* c_in, const float* c_out,
                                     const int64_t* q_len, int t, float* dh, float* dc, float* dgates, int B, int H) {
  const int64_t total = (int64_t)B * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / H), j = (int)(i - (int64_t)b * H);
    const int64_t g0 = (int64_t)b * 4 * H + j;
    float di = 0.f, df = 0.f, dg = 0.f, dgo = 0.f;
    if ((int64_t)t < q_len[b]) {
      const float gi = gates[g0], gf = gates[g0 + H], gg = gates[g0 + 2 * H], go = gates[g0 + 3 * H];
      const float tc = tanhf(c_out[i]);
      const float dhv = dh[i];
      const float dct = dc[i] + dhv * go * (1.f - tc * tc);
      di = dct * gg * gi * (1.f - gi);
      df = dct * c_in[i] * gf * (1.f - gf);
      dg = dct * gi * (1.f - gg * gg);
      dgo = dhv * tc * go * (1.f - go);
      dc[i] = dct * gf;
      dh[i] = 0.f;
    }
    dgates[g0] = di; dgates[g0 + H] = df; dgates[g0 + 2 * H] = dg; dgates[g0 + 3 * H] = dgo;
  }
}

// ------------------------------------------------------------------ attention score (x_conv)
// x = relu(v' (+|*) q') is xs [M][mid]; for do_option '|' x = relu(cat[v', tile(q')]) has 2*mid channels:
// the v' half is xs, the q' half is relu(qcat[b][:]) replicated over positions (but with its own
// per-position dropout mask, as nn.Dropout acts on the concatenated tensor).  xld = channels of x.
// XB: xs holds bf16 (the bf16 path stores x = relu(v' (+|*) q') as bf16)
template <int G, bool XB>
__global__ void att_score_fwd_kernel(const void* xs, const float* wx, int wx_ld, const float* bx, float* score,
                                     int64_t M, int P, int mid, float p, float inv_keep, uint64_t seed,
                                     const float* qcat) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int nch = mid >> 2;
  const int xld = qcat ? 2 * mid : mid;
  for (int64_t m = wave; m < M; m += nwaves) {
    const int64_t b = m / P;
    const int pp = (int)(m - b * P);
    const char* row = static_cast<const char*>(xs) + m * mid * (XB ? 2 : 4);
    float acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = 0.f;
    for (int c = lane; c < nch; c += 64) {
      float4 x = ld4<XB>(row, c);
      if (p > 0.f) {
        const uint64_t e = (uint64_t)m * xld + 4 * c;
        { const float4 ds_ = drop_scale4(seed, e, p, inv_keep); x.x *= ds_.x; x.y *= ds_.y; x.z *= ds_.z; x.w *= ds_.w; }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float4 w = reinterpret_cast<const float4*>(wx + (int64_t)g * wx_ld)[c];
        acc[g] += x.x * w.x + x.y * w.y + x.z * w.z + x.w * w.w;
      }
      if (qcat) {
        float4 q = reinterpret_cast<const float4*>(qcat + b * mid)[c];
        q.x = fmaxf(q.x, 0.f); q.y = fmaxf(q.y, 0.f); q.z = fmaxf(q.z, 0.f); q.w = fmaxf(q.w, 0.f);
        if (p > 0.f) {
          const uint64_t e = (uint64_t)m * xld + mid + 4 * c;
          { const float4 ds_ = drop_scale4(seed, e, p, inv_keep); q.x *= ds_.x; q.y *= ds_.y; q.z *= ds_.z; q.w *= ds_.w; }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float4 w = reinterpret_cast<const float4*>(wx + (int64_t)g * wx_ld + mid)[c];
          acc[g] += q.x * w.x + q.y * w.y + q.z * w.z + q.w * w.w;
        }
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float v = wave_sum(acc[g]);
      if (lane == 0) score[(b * G + g) * P + pp] = v + bx[g];
    }
  }
}

// fp32 x without the concatenated half, mid a multiple of 256 up to 1024 (the reference's 1024): the general kernel above walks
// a row in mid/256 DEPENDENT rounds of 16-byte loads with one row per wave in flight and re-reads the x_conv weights for every
// row (3.9 TB/s).  Here a lane issues all IT loads of a row before it touches the first value, the next row's loads are issued
// before the current row is reduced, and the weights live in registers.  Same per-lane operation order as the general kernel.
template <int G, int IT>
__global__ __launch_bounds__(256) void att_score_fwd_rows_kernel(const float* xs, const float* wx, int wx_ld, const float* bx,
                                                                 float* score, int64_t M, int P, float p, float inv_keep,
                                                                 uint64_t seed) {
  constexpr int mid = 256 * IT;
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float4 w[IT][G];
#pragma unroll
  for (int i = 0; i < IT; ++i)
#pragma unroll
    for (int g = 0; g < G; ++g) w[i][g] = reinterpret_cast<const float4*>(wx + (int64_t)g * wx_ld)[lane + 64 * i];
  float bias[G];
#pragma unroll
  for (int g = 0; g < G; ++g) bias[g] = bx[g];
  float4 cur[IT], nxt[IT];
  int64_t m = wave;
  if (m < M) {
#pragma unroll
    for (int i = 0; i < IT; ++i) cur[i] = reinterpret_cast<const float4*>(xs + m * mid)[lane + 64 * i];
  }
  for (; m < M; m += nwaves) {
    const int64_t mn = m + nwaves;
    if (mn < M) {
#pragma unroll
      for (int i = 0; i < IT; ++i) nxt[i] = reinterpret_cast<const float4*>(xs + mn * mid)[lane + 64 * i];
    }
    float acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      float4 x = cur[i];
      if (p > 0.f) {
        const uint64_t e = (uint64_t)m * mid + 4 * (lane + 64 * i);
        const float4 ds_ = drop_scale4(seed, e, p, inv_keep);
        x.x *= ds_.x; x.y *= ds_.y; x.z *= ds_.z; x.w *= ds_.w;
      }
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] += x.x * w[i][g].x + x.y * w[i][g].y + x.z * w[i][g].z + x.w * w[i][g].w;
    }
    const int64_t b = m / P;
    const int pp = (int)(m - b * P);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float v = wave_sum(acc[g]);
      if (lane == 0) score[(b * G + g) * P + pp] = v + bias[g];
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) cur[i] = nxt[i];
  }
}

// The bf16 path's form of the above (x = relu(v' (+|*) q') stored as bf16, no concatenated half, mid <= 1024): the generic
// kernel walked a row in four dependent rounds of 8-byte loads and re-read the x_conv weights for every row -- 2.3 TB/s.  Here
// a lane owns 8 channels per round (16-byte loads), keeps its weights in registers for the whole kernel, and a wave has the
// loads of TWO rows in flight before it touches the first value.
template <int G>
__global__ __launch_bounds__(256) void att_score_fwd_bf16_kernel(const uint16_t* xs, const float* wx, int wx_ld, const float* bx,
                                                                 float* score, int64_t M, int P, int mid, float p, float inv_keep,
                                                                 uint64_t seed) {
  constexpr int IT = 2;                                   // rounds of 64 lanes x 8 channels: mid <= 1024
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int n8 = mid >> 3;
  float w[IT][G][8];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c8 = lane + 64 * i;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int k = 0; k < 8; ++k) w[i][g][k] = c8 < n8 ? wx[(int64_t)g * wx_ld + 8 * c8 + k] : 0.f;
  }
  float bias[G];
#pragma unroll
  for (int g = 0; g < G; ++g) bias[g] = bx[g];
  for (int64_t m0 = 2 * wave; m0 < M; m0 += 2 * nwaves) {
    uint4 xr[2][IT];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int c8 = lane + 64 * i;
        xr[rr][i] = (m0 + rr < M && c8 < n8) ? *reinterpret_cast<const uint4*>(xs + (m0 + rr) * mid + 8 * c8) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int64_t m = m0 + rr;
      float acc[G];
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = 0.f;
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int c8 = lane + 64 * i;
        const uint32_t u[4] = {xr[rr][i].x, xr[rr][i].y, xr[rr][i].z, xr[rr][i].w};
        float x[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { x[2 * k] = __uint_as_float(u[k] << 16); x[2 * k + 1] = __uint_as_float(u[k] & 0xffff0000u); }
        if (p > 0.f) {
          const uint64_t e = (uint64_t)m * mid + 8 * c8;
          const float4 s0 = drop_scale4(seed, e, p, inv_keep), s1 = drop_scale4(seed, e + 4, p, inv_keep);
          x[0] *= s0.x; x[1] *= s0.y; x[2] *= s0.z; x[3] *= s0.w; x[4] *= s1.x; x[5] *= s1.y; x[6] *= s1.z; x[7] *= s1.w;
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[g] += x[k] * w[i][g][k];
      }
      if (m < M) {
        const int64_t b = m / P;
        const int pp = (int)(m - b * P);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float v = wave_sum(acc[g]);
          if (lane == 0) score[(b * G + g) * P + pp] = v + bias[g];
        }
      }
    }
  }
}

// grid (B, RS); thread -> float4 column chunks; loops the rows of its split.
// mode 0 '+': xs <- dz = (x>0) * mask * sum_g ds*wx            dq' part = sum_p dz
// mode 1 '*': xs <- dv' = dz * q'[b]                             dq' part = sum_p dz * v'
// mode 2 '|': xs <- dv' = dz (v' half);  q' half: dq' part = (q'>0) * sum_p mask_q * sum_g ds*wx[g][mid+n]
// dwx_part[part][G][xld]: sum_p ds[g] * dropout(x) over the rows of the part (both halves for '|').
template <int G, bool XB>
__global__ void att_score_bwd_kernel(const float* dscore, const float* wx, int wx_ld, void* xs, float* dwx_part,
                                     float* dq_part, int P, int mid, int RS, float p, float inv_keep, uint64_t seed,
                                     int mode, const float* vprime, const float* qp) {
  const int b = blockIdx.x, rs = blockIdx.y;
  const int rows_per = (P + RS - 1) / RS;
  const int p0 = rs * rows_per, p1 = min(P, p0 + rows_per);
  const int nch = mid >> 2;
  const int xld = mode == 2 ? 2 * mid : mid;
  const int64_t part = (int64_t)b * RS + rs;
  for (int c = threadIdx.x; c < nch; c += blockDim.x) {
    float4 w[G], dw[G], w2[G], dw2[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      w[g] = reinterpret_cast<const float4*>(wx + (int64_t)g * wx_ld)[c];
      dw[g] = make_float4(0.f, 0.f, 0.f, 0.f);
      dw2[g] = make_float4(0.f, 0.f, 0.f, 0.f);
      w2[g] = mode == 2 ? reinterpret_cast<const float4*>(wx + (int64_t)g * wx_ld + mid)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mode != 0) qv = reinterpret_cast<const float4*>(qp + (int64_t)b * mid)[c];
    const float4 rq = make_float4(fmaxf(qv.x, 0.f), fmaxf(qv.y, 0.f), fmaxf(qv.z, 0.f), fmaxf(qv.w, 0.f));
    float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
    // rows in batches of four: the four loads of x (and the batch's dscore values) are issued before the first of them is
    // used -- xs is rewritten in place, so hipcc may not move a row's load above the previous row's store by itself, and one
    // dependent load -> compute -> store chain per row left the kernel at 4 TB/s
    for (int pb = p0; pb < p1; pb += 4) {
      float4 xb[4];
      float dsb[4][G];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pp = pb + u < p1 ? pb + u : p1 - 1;
        const int64_t m = (int64_t)b * P + pp;
        xb[u] = ld4<XB>(static_cast<const char*>(xs) + m * mid * (XB ? 2 : 4), c);
#pragma unroll
        for (int g = 0; g < G; ++g) dsb[u][g] = dscore[((int64_t)b * G + g) * P + pp];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
      const int pp = pb + u;
      if (pp >= p1) break;
      const int64_t m = (int64_t)b * P + pp;
      char* xrow = static_cast<char*>(xs) + m * mid * (XB ? 2 : 4);
      const float4 x = xb[u];
      float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
      if (p > 0.f) {
        const uint64_t e = (uint64_t)m * xld + 4 * c;
        sc = drop_scale4(seed, e, p, inv_keep);
      }
      float ds[G];
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        ds[g] = dsb[u][g];
        t.x += ds[g] * w[g].x; t.y += ds[g] * w[g].y; t.z += ds[g] * w[g].z; t.w += ds[g] * w[g].w;
        dw[g].x += ds[g] * x.x * sc.x; dw[g].y += ds[g] * x.y * sc.y; dw[g].z += ds[g] * x.z * sc.z; dw[g].w += ds[g] * x.w * sc.w;
      }
      float4 d;
      d.x = x.x > 0.f ? t.x * sc.x : 0.f; d.y = x.y > 0.f ? t.y * sc.y : 0.f;
      d.z = x.z > 0.f ? t.z * sc.z : 0.f; d.w = x.w > 0.f ? t.w * sc.w : 0.f;
      if (mode == 0) {
        dq.x += d.x; dq.y += d.y; dq.z += d.z; dq.w += d.w;
      } else if (mode == 1) {
        const float4 vp = reinterpret_cast<const float4*>(vprime + m * mid)[c];
        dq.x += d.x * vp.x; dq.y += d.y * vp.y; dq.z += d.z * vp.z; dq.w += d.w * vp.w;
        d.x *= qv.x; d.y *= qv.y; d.z *= qv.z; d.w *= qv.w;
      } else {
        float4 s2 = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p > 0.f) {
          const uint64_t e = (uint64_t)m * xld + mid + 4 * c;
          s2 = drop_scale4(seed, e, p, inv_keep);
        }
        float4 t2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          t2.x += ds[g] * w2[g].x; t2.y += ds[g] * w2[g].y; t2.z += ds[g] * w2[g].z; t2.w += ds[g] * w2[g].w;
          dw2[g].x += ds[g] * rq.x * s2.x; dw2[g].y += ds[g] * rq.y * s2.y; dw2[g].z += ds[g] * rq.z * s2.z; dw2[g].w += ds[g] * rq.w * s2.w;
        }
        dq.x += qv.x > 0.f ? t2.x * s2.x : 0.f; dq.y += qv.y > 0.f ? t2.y * s2.y : 0.f;
        dq.z += qv.z > 0.f ? t2.z * s2.z : 0.f; dq.w += qv.w > 0.f ? t2.w * s2.w : 0.f;
      }
      st4<XB>(xrow, c, d);
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      reinterpret_cast<float4*>(dwx_part + (part * G + g) * xld)[c] = dw[g];
      if (mode == 2) reinterpret_cast<float4*>(dwx_part + (part * G + g) * xld + mid)[c] = dw2[g];
    }
    reinterpret_cast<float4*>(dq_part + part * mid)[c] = dq;
  }
}

// ------------------------------------------------------------------ softmax over positions + weighted sum
__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
  return r;
}

// Softmax over the P positions of every glimpse into LDS (and to `probs` from the first channel block).
template <int G>
__device__ __forceinline__ void att_softmax_to_lds(const float* score, float* probs, float* pr, float* red, int b, int P,
                                                   bool write) {
  const int tid = threadIdx.x;
  for (int g = 0; g < G; ++g) {
    const float* s = score + ((int64_t)b * G + g) * P;
    float mx = -INFINITY;
    for (int i = tid; i < P; i += blockDim.x) mx = fmaxf(mx, s[i]);
    mx = block_reduce(mx, red, true);
    float sum = 0.f;
    for (int i = tid; i < P; i += blockDim.x) { const float e = expf(s[i] - mx); pr[g * P + i] = e; sum += e; }
    sum = block_reduce(sum, red, false);
    const float inv = 1.f / sum;
    for (int i = tid; i < P; i += blockDim.x) {
      const float v = pr[g * P + i] * inv;
      pr[g * P + i] = v;
      if (write) probs[((int64_t)b * G + g) * P + i] = v;
    }
  }
}

// C % 4 == 0: 256 threads = 16 position groups x 16 channel quads, 16-byte loads, four positions in flight per thread
// (the scalar form below runs at 26 % of the HBM rate: one 4-byte load per FMA pair, 169 dependent iterations).
// grid (B, ceil(C/64)); dynamic LDS: G*P + 16 + 16*G*64 floats.  Sum order per channel: positions pg, pg+16, ... within
// a group, then the 16 groups in order.
template <int G>
__global__ void att_apply_fwd_v4_kernel(const float* score, const float* vn, float* probs, float* out, int64_t out_ld,
                                        int P, int C) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* pr = sm;            // [G][P]
  float* red = sm + G * P;   // [16]
  float* part = red + 16;    // [16][G][64]
  const int b = blockIdx.x, tid = threadIdx.x;
  att_softmax_to_lds<G>(score, probs, pr, red, b, P, blockIdx.y == 0);
  __syncthreads();
  const int cq = tid & 15, pg = tid >> 4;
  const int c = blockIdx.y * 64 + cq * 4;
  float4 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C) {
    const float* vb = vn + (int64_t)b * P * C + c;
    int i = pg;
    for (; i + 48 < P; i += 64) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(vb + (int64_t)(i + 16 * u) * C);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float w = pr[g * P + i + 16 * u];
          acc[g].x += w * v[u].x; acc[g].y += w * v[u].y; acc[g].z += w * v[u].z; acc[g].w += w * v[u].w;
        }
    }
    for (; i < P; i += 16) {
      const float4 v = *reinterpret_cast<const float4*>(vb + (int64_t)i * C);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float w = pr[g * P + i];
        acc[g].x += w * v.x; acc[g].y += w * v.y; acc[g].z += w * v.z; acc[g].w += w * v.w;
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) *reinterpret_cast<float4*>(part + (pg * G + g) * 64 + cq * 4) = acc[g];
  __syncthreads();
  // 64 * G outputs, one per thread (G <= 4)
  for (int o = tid; o < 64 * G; o += 256) {
    const int g = o >> 6, cl = o & 63;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += part[(k * G + g) * 64 + cl];
    if (blockIdx.y * 64 + cl < C) out[(int64_t)b * out_ld + g * C + blockIdx.y * 64 + cl] = v;
  }
}

// grid (B, ceil(C/64)), 256 threads = 4 position groups x 64 channels; dynamic LDS: G*P + 16 + 4*G*64 floats
template <int G>
__global__ void att_apply_fwd_kernel(const float* score, const float* vn, float* probs, float* out,
                                     int64_t out_ld, int P, int C) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* pr = sm;            // [G][P]
  float* red = sm + G * P;   // [16]
  float* part = red + 16;    // [4][G][64]
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int g = 0; g < G; ++g) {
    const float* s = score + ((int64_t)b * G + g) * P;
    float mx = -INFINITY;
    for (int i = tid; i < P; i += blockDim.x) mx = fmaxf(mx, s[i]);
    mx = block_reduce(mx, red, true);
    float sum = 0.f;
    for (int i = tid; i < P; i += blockDim.x) { const float e = expf(s[i] - mx); pr[g * P + i] = e; sum += e; }
    sum = block_reduce(sum, red, false);
    const float inv = 1.f / sum;
    for (int i = tid; i < P; i += blockDim.x) {
      const float v = pr[g * P + i] * inv;
      pr[g * P + i] = v;
      if (blockIdx.y == 0) probs[((int64_t)b * G + g) * P + i] = v;
    }
  }
  __syncthreads();
  const int cl = tid & 63, pg = tid >> 6;
  const int c = blockIdx.y * 64 + cl;
  float acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = 0.f;
  if (c < C) {
    const float* vb = vn + (int64_t)b * P * C + c;
    for (int i = pg; i < P; i += 4) {
      const float v = vb[(int64_t)i * C];
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] += pr[g * P + i] * v;
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) part[(pg * G + g) * 64 + cl] = acc[g];
  __syncthreads();
  if (pg == 0 && c < C) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float v = part[g * 64 + cl] + part[(G + g) * 64 + cl] + part[(2 * G + g) * 64 + cl] + part[(3 * G + g) * 64 + cl];
      out[(int64_t)b * out_ld + g * C + c] = v;
    }
  }
}

// pass 1: one wave per (b,p): dprob -> dscore buffer, dvn row written
template <int G>
__global__ void att_apply_bwd_rows_kernel(const float* dout, int64_t dout_ld, const float* probs, const float* vn,
                                          float* dprob, float* dvn, int64_t M, int P, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int nch = C >> 2;
  for (int64_t m = wave; m < M; m += nwaves) {
    const int64_t b = m / P;
    const int pp = (int)(m - b * P);
    float pr[G], acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { pr[g] = probs[(b * G + g) * P + pp]; acc[g] = 0.f; }
    const float4* v = reinterpret_cast<const float4*>(vn + m * C);
    float4* d = reinterpret_cast<float4*>(dvn + m * C);
    for (int c = lane; c < nch; c += 64) {
      const float4 x = v[c];
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float4 go = reinterpret_cast<const float4*>(dout + b * dout_ld + (int64_t)g * C)[c];
        acc[g] += x.x * go.x + x.y * go.y + x.z * go.z + x.w * go.w;
        o.x += pr[g] * go.x; o.y += pr[g] * go.y; o.z += pr[g] * go.z; o.w += pr[g] * go.w;
      }
      if (dvn) d[c] = o;       // dvn == null: vqa_l2norm_bwd_joined recomputes this branch where it is consumed
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float s = wave_sum(acc[g]);
      if (lane == 0) dprob[(b * G + g) * P + pp] = s;
    }
  }
}
// pass 2: block per (b,g): dscore = probs * (dprob - sum_p probs*dprob), in place on dprob
// rowsum (optional) [B*G]: sum_p dscore of the block's row -- the per-sample part of the x_conv bias gradient
__global__ void softmax_bwd_kernel(const float* probs, float* dscore, int P, float* rowsum) {
  __shared__ float red[16];
  const int64_t base = (int64_t)blockIdx.x * P;
  float s = 0.f;
  for (int i = threadIdx.x; i < P; i += blockDim.x) s += probs[base + i] * dscore[base + i];
  s = block_reduce(s, red, false);
  float t = 0.f;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    const float d = probs[base + i] * (dscore[base + i] - s);
    dscore[base + i] = d;
    t += d;
  }
  if (rowsum) {
    t = block_reduce(t, red, false);
    if (threadIdx.x == 0) rowsum[blockIdx.x] = t;
  }
}

// ------------------------------------------------------------------ soft-target CE + VQA score
__global__ void softce_kernel(const float* logits, int64_t ld, const int64_t* a_idx, const int64_t* a_val, int kmax,
                              int A, float inv_batch, float* loss_rows, float* score_rows, float* dlogits, int64_t dld) {
  __shared__ float red[16];
  __shared__ int redi[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* row = logits + (int64_t)b * ld;
  float mx = -INFINITY;
  int am = 0x7fffffff;
  for (int i = tid; i < A; i += blockDim.x) { const float v = row[i]; if (v > mx) { mx = v; am = i; } }
  // arg-max with smallest-index tie break (torch.max returns the first maximal element)
  {
    const int lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(mx, o, 64);
      const int oi = __shfl_xor(am, o, 64);
      if (ov > mx || (ov == mx && oi < am)) { mx = ov; am = oi; }
    }
    if (lane == 0) { red[w] = mx; redi[w] = am; }
    __syncthreads();
    mx = red[0]; am = redi[0];
    for (int i = 1; i < nw; ++i) if (red[i] > mx || (red[i] == mx && redi[i] < am)) { mx = red[i]; am = redi[i]; }
  }
  float sum = 0.f;
  for (int i = tid; i < A; i += blockDim.x) sum += expf(row[i] - mx);
  sum = block_reduce(sum, red, false);
  const float lse = mx + logf(sum);
  float wsum = 0.f, loss = 0.f, agree = 0.f;
  for (int k = 0; k < kmax; ++k) {
    const int64_t idx = a_idx[(int64_t)b * kmax + k];
    if (idx <= 0 || idx > A) continue;
    const float w = (float)a_val[(int64_t)b * kmax + k] / 10.0f;
    wsum += w;
    loss += w * (lse - row[idx - 1]);
    if ((int)(idx - 1) == am) agree = (float)a_val[(int64_t)b * kmax + k];
  }
  if (tid == 0) {
    loss_rows[b] = loss * inv_batch;
    score_rows[b] = fminf(agree * 0.3f, 1.0f);
  }
  if (dlogits) {
    float* d = dlogits + (int64_t)b * dld;
    const float invs = 1.f / sum;
    for (int i = tid; i < A; i += blockDim.x) {
      float g = wsum * expf(row[i] - mx) * invs;
      for (int k = 0; k < kmax; ++k)
        if (a_idx[(int64_t)b * kmax + k] == (int64_t)i + 1) g -= (float)a_val[(int64_t)b * kmax + k] / 10.0f;
      d[i] = g * inv_batch;
    }
  }
}

// ------------------------------------------------------------------ column sums (two stage, deterministic)
// stage 1: grid (ceil(cols/64), splits); 256 threads = 4 row lanes x 64 columns
__global__ void colsum_stage1(const float* x, int64_t ld, const uint8_t* mask, int64_t rows, int cols,
                              int64_t rows_per, float* slab) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per;
  const int64_t r1 = r0 + rows_per < rows ? r0 + rows_per : rows;
  float acc = 0.f;
  if (c < cols)
    for (int64_t r = r0 + rl; r < r1; r += 4) {
      const float v = x[r * ld + c];
      if (!mask || mask[r * cols + c] != 4) acc += v;
    }
  part[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < cols) slab[(int64_t)blockIdx.y * cols + c] = part[0][cl] + part[1][cl] + part[2][cl] + part[3][cl];
}
__global__ void colsum_stage2(const float* slab, int splits, int cols, float* out, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float v = 0.f;
  for (int s = 0; s < splits; ++s) v += slab[(int64_t)s * cols + c];
  out[c] = accumulate ? out[c] + v : v;
}

static void colsum_plan(int64_t rows, int cols, int* splits, int64_t* rows_per) {
  const int ctiles = (cols + 63) / 64;
  int64_t s = (1024 + ctiles - 1) / ctiles;          // aim for ~1024 workgroups
  const int64_t max_s = (rows + 63) / 64;            // at least 64 rows per split
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  *rows_per = (rows + s - 1) / s;
  *splits = (int)((rows + *rows_per - 1) / *rows_per);
}
int64_t colsum_ws_bytes(int64_t rows, int cols) {
  int splits; int64_t rp;
  colsum_plan(rows, cols, &splits, &rp);
  return (int64_t)splits * cols * 4;
}
int colsum_launch(const float* x, int64_t ld, const uint8_t* mask, int64_t rows, int cols, float* out,
                  int accumulate, float* ws, int64_t ws_bytes, hipStream_t s) {
  int splits; int64_t rp;
  colsum_plan(rows, cols, &splits, &rp);
  if (!ws || ws_bytes < (int64_t)splits * cols * 4) {
    set_error("colsum: workspace %lld < %lld", (long long)ws_bytes, (long long)splits * cols * 4);
    return VQA_ERR_WORKSPACE;
  }
  hipLaunchKernelGGL(colsum_stage1, dim3((cols + 63) / 64, splits), dim3(256), 0, s, x, ld, mask, rows, cols, rp, ws);
  int rc = check_hip(hipGetLastError(), "colsum_stage1 launch");
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_stage2, dim3((cols + 255) / 256), dim3(256), 0, s, ws, splits, cols, out, accumulate);
  return check_hip(hipGetLastError(), "colsum_stage2 launch");
}

// out[g] = sum_{b,p} x[b][g][p], deterministic two-stage sum inside one launch-pair-free kernel: grid (G), 1024
// threads; thread groups of 64 lanes take whole rows [b][g][:] (coalesced), partial sums combine in a fixed order.
// (The result is the x_conv bias gradient, ~0 by construction: softmax is shift invariant.)
__global__ __launch_bounds__(1024) void sum_bgp_kernel(const float* x, float* out, int B, int G, int P) {
  __shared__ float red[16];
  const int g = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;     // 16 waves, wave w takes samples w, w+16, ...
  float s = 0.f;
  for (int b = w; b < B; b += 16) {
    const float* row = x + ((int64_t)b * G + g) * P;
    for (int i = lane; i < P; i += 64) s += row[i];
  }
  s = block_reduce(s, red, false);
  if (threadIdx.x == 0) out[g] = s;
}

// out[b][n] = sum_r part[(b*parts + r)*cols + n]
__global__ void sum_parts_kernel(const float* part, float* out, int parts, int cols) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float v = 0.f;
  for (int r = 0; r < parts; ++r) v += part[((int64_t)b * parts + r) * cols + c];
  out[(int64_t)b * cols + c] = v;
}

__global__ void relu_drop_bwd_kernel(const float* y, const float* dy, float* dx, int64_t n, float p, float inv_keep,
                                     uint64_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float g = y[i] > 0.f ? dy[i] : 0.f;
    if (p > 0.f) g *= drop_scale(seed, (uint64_t)i, p, inv_keep);
    dx[i] = g;
  }
}

__global__ void add2d_kernel(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy,
                             int64_t rows, int cols) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    y[r * ldy + c] = a[r * lda + c] + (b ? b[r * ldb + c] : 0.f);
  }
}

// fp16 -> fp32, 8 values (16 bytes in, 32 bytes out) per thread and iteration
__global__ void half_to_float_kernel(const __half* x, float* y, int64_t n) {
  const int64_t n8 = n / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 raw = reinterpret_cast<const uint4*>(x)[i];
    const __half2* h = reinterpret_cast<const __half2*>(&raw);
    const float2 a = __half22float2(h[0]), b = __half22float2(h[1]), c = __half22float2(h[2]), d = __half22float2(h[3]);
    reinterpret_cast<float4*>(y)[2 * i] = make_float4(a.x, a.y, b.x, b.y);
    reinterpret_cast<float4*>(y)[2 * i + 1] = make_float4(c.x, c.y, d.x, d.y);
  }
  if (blockIdx.x == 0 && threadIdx.x < n - 8 * n8) y[8 * n8 + threadIdx.x] = __half2float(x[8 * n8 + threadIdx.x]);
}

__global__ void scale_by_kernel(float* x, int64_t n, const float* s) {
  const float k = *s;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= k;
}

__global__ void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, float step_size, float beta1,
                            float beta2, float eps, float inv_sqrt_bc2, float grad_scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gr = g[i] * grad_scale;
    const float mi = beta1 * m[i] + (1.f - beta1) * gr;
    const float vi = beta2 * v[i] + (1.f - beta2) * gr * gr;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] -= step_size * (mi / denom);
  }
}

}  // namespace vqa

using namespace vqa;

#define STREAM ((hipStream_t)stream)
#define KEEP(p) ((p) > 0.f ? 1.0f / (1.0f - (p)) : 1.0f)

template <int G>
static void l2norm_bwd_launch(const float* dvn, const float* vn, const float* norm, void* dpooled, int mode, int64_t rows,
                              int positions, int C, float p, uint64_t seed, const float* dout, int64_t dout_ld,
                              const float* probs, const float* dv_in, float p_v, uint64_t seed_v, hipStream_t s) {
#define L2B(OB)                                                                                                            \
  hipLaunchKernelGGL((l2norm_bwd_kernel<OB, G>), dim3(grid_for(rows, 4)), dim3(256), 0, s, dvn, vn, norm, dpooled, rows, C, p, \
                     KEEP(p), seed, positions, dout, dout_ld, probs, dv_in, p_v, KEEP(p_v), seed_v)
  if (mode == 2) L2B(2); else if (mode == 1) L2B(1); else L2B(0);
#undef L2B
}

extern "C" {

int vqa_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, vqa_stream_t stream) {
  set_launch_tag(n >= (1 << 22) ? 1 : 0);       // 1 = a large tensor (the attention dropout on v), 0 = the small sites
  ProfScope prof(VQA_K_DROPOUT, (hipStream_t)stream);
  VQA_REQUIRE(x && y && n >= 0 && p >= 0.f && p < 1.f, "vqa_dropout: bad args");
  if (n == 0) return VQA_OK;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n, 256)), dim3(256), 0, STREAM, x, y, n, p, KEEP(p), seed);
  return check_hip(hipGetLastError(), "dropout launch");
}

int vqa_dropout_add(const float* x, float* y, int64_t n, float p, uint64_t seed, vqa_stream_t stream) {
  set_launch_tag(n >= (1 << 22) ? 1 : 0);
  ProfScope prof(VQA_K_DROPOUT, (hipStream_t)stream);
  VQA_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, "vqa_dropout_add: bad args");
  VQA_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, "vqa_dropout_add: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(dropout_add_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, STREAM, x, y, n, p, KEEP(p), seed);
  return check_hip(hipGetLastError(), "dropout_add launch");
}

int vqa_l2norm_fwd(const float* pooled, float* vn, float* norm, int64_t rows, int C, float p, uint64_t seed,
                   void* vdrop, int vdrop_is_bf16, float p2, uint64_t seed2, vqa_stream_t stream) {
  set_launch_tag(-1);
  ProfScope prof(VQA_K_L2NORM_FWD, (hipStream_t)stream);
  VQA_REQUIRE(pooled && vn && norm && rows > 0 && C > 0 && C % 4 == 0, "vqa_l2norm_fwd: bad args (C=%d)", C);
  VQA_REQUIRE(p2 >= 0.f && p2 < 1.f, "vqa_l2norm_fwd: p2 must be in [0, 1)");
  if (vdrop && vdrop_is_bf16)
    hipLaunchKernelGGL(l2norm_fwd_kernel<true>, dim3(grid_for(rows, 4)), dim3(256), 0, STREAM, pooled, vn, norm, rows, C, p,
                       KEEP(p), seed, vdrop, p2, KEEP(p2), seed2);
  else
    hipLaunchKernelGGL(l2norm_fwd_kernel<false>, dim3(grid_for(rows, 4)), dim3(256), 0, STREAM, pooled, vn, norm, rows, C, p,
                       KEEP(p), seed, vdrop, p2, KEEP(p2), seed2);
  return check_hip(hipGetLastError(), "l2norm_fwd launch");
}

#define DISPATCH_G(G, ...)                                         \
  switch (G) {                                                     \
    case 1: { constexpr int kG = 1; __VA_ARGS__; } break;          \
    case 2: { constexpr int kG = 2; __VA_ARGS__; } break;          \
    case 3: { constexpr int kG = 3; __VA_ARGS__; } break;          \
    case 4: { constexpr int kG = 4; __VA_ARGS__; } break;          \
    default: set_error("glimpses=%d unsupported (1..4)", G); return VQA_ERR_INVALID; \
  }

int vqa_l2norm_bwd(const float* dvn, const float* vn, const float* norm, void* dpooled, int dpooled_mode, int64_t rows,
                   int positions, int C, float p, uint64_t seed, vqa_stream_t stream) {
  set_launch_tag(-1);
  ProfScope prof(VQA_K_L2NORM_BWD, (hipStream_t)stream);
  VQA_REQUIRE(dvn && vn && norm && dpooled && rows > 0 && C % 4 == 0 && dpooled_mode >= 0 && dpooled_mode <= 2, "vqa_l2norm_bwd: bad args");
  VQA_REQUIRE(dpooled_mode != 2 || (C % 16 == 0 && positions > 0 && rows % positions == 0),
              "vqa_l2norm_bwd: the channel-blocked output needs C %% 16 == 0 and rows = images x positions (C=%d, positions=%d)", C, positions);
  l2norm_bwd_launch<0>(dvn, vn, norm, dpooled, dpooled_mode, rows, positions > 0 ? positions : 1, C, p, seed, nullptr, 0, nullptr,
                       nullptr, 0.f, 0, STREAM);
  return check_hip(hipGetLastError(), "l2norm_bwd launch");
}

int vqa_l2norm_bwd_joined(const float* dout, int64_t dout_ld, const float* probs, int G, const float* dv_in, float p_v,
                          uint64_t seed_v, const float* vn, const float* norm, void* dpooled, int dpooled_mode, int64_t rows,
                          int positions, int C, float p, uint64_t seed, vqa_stream_t stream) {
  set_launch_tag(-1);
  ProfScope prof(VQA_K_L2NORM_BWD, (hipStream_t)stream);
  VQA_REQUIRE(dout && probs && dv_in && vn && norm && dpooled && rows > 0 && C % 4 == 0 && dout_ld % 4 == 0 && dpooled_mode >= 0 &&
                  dpooled_mode <= 2 && positions > 0 && rows % positions == 0, "vqa_l2norm_bwd_joined: bad args");
  VQA_REQUIRE(dpooled_mode != 2 || C % 16 == 0, "vqa_l2norm_bwd_joined: the channel-blocked output needs C %% 16 == 0 (C=%d)", C);
  DISPATCH_G(G, l2norm_bwd_launch<kG>(nullptr, vn, norm, dpooled, dpooled_mode, rows, positions, C, p, seed, dout, dout_ld, probs,
                                      dv_in, p_v, seed_v, STREAM));
  return check_hip(hipGetLastError(), "l2norm_bwd_joined launch");
}

int vqa_embed_tanh_fwd(const int64_t* q, const float* emb, float* x, int B, int T, int E, int V, float p,
                       uint64_t seed, int32_t* bad_tokens, vqa_stream_t stream) {
  VQA_REQUIRE(q && emb && x && B > 0 && T > 0 && E > 0 && V > 0, "vqa_embed_tanh_fwd: bad args");
  hipLaunchKernelGGL(embed_tanh_fwd_kernel, dim3(grid_for((int64_t)B * T * E, 256)), dim3(256), 0, STREAM, q, emb, x,
                     B, T, E, V, p, KEEP(p), seed, bad_tokens);
  return check_hip(hipGetLastError(), "embed_tanh_fwd launch");
}

int64_t vqa_embed_tanh_bwd_workspace_bytes(int B, int T, int V) {
  if (B <= 0 || T <= 0 || V <= 0) return 0;
  return ((int64_t)3 * V + 2 + (int64_t)B * T) * 4;             // counts[V] offsets[V+1] cursor[V] slots[B*T] (int32)
}

int vqa_embed_tanh_bwd(const int64_t* q, const float* x, const float* dx, float* demb, int B, int T, int E, int V,
                       float p, uint64_t seed, void* workspace, int64_t workspace_bytes, vqa_stream_t stream) {
  VQA_REQUIRE(q && x && dx && demb && B > 0 && T > 0 && E > 0 && V > 0, "vqa_embed_tanh_bwd: bad args");
  if (!workspace) {                                              // no workspace: the scanning kernel, O(V * B*T)
    hipLaunchKernelGGL(embed_tanh_bwd_kernel, dim3(V), dim3(256), 0, STREAM, q, x, dx, demb, B, T, E, V, p, KEEP(p), seed);
    return check_hip(hipGetLastError(), "embed_tanh_bwd launch");
  }
  if (workspace_bytes < vqa_embed_tanh_bwd_workspace_bytes(B, T, V)) {
    set_error("vqa_embed_tanh_bwd: workspace %lld < %lld", (long long)workspace_bytes,
              (long long)vqa_embed_tanh_bwd_workspace_bytes(B, T, V));
    return VQA_ERR_WORKSPACE;
  }
  const int N = B * T;
  int* const counts = static_cast<int*>(workspace);
  int* const offsets = counts + V;
  int* const cursor = offsets + V + 1;
  int* const slots = cursor + V;
  int rc = check_hip(hipMemsetAsync(counts, 0, (size_t)V * 4, STREAM), "embed_tanh_bwd memset");
  if (rc) return rc;
  hipLaunchKernelGGL(embed_hist_kernel, dim3((N + 255) / 256), dim3(256), 0, STREAM, q, N, V, counts);
  hipLaunchKernelGGL(embed_scan_kernel, dim3(1), dim3(1024), 0, STREAM, counts, offsets, cursor, V);
  hipLaunchKernelGGL(embed_fill_kernel, dim3((N + 255) / 256), dim3(256), 0, STREAM, q, N, V, cursor, slots);
  hipLaunchKernelGGL(embed_tanh_bwd_binned_kernel, dim3(V), dim3(256), 0, STREAM, q, x, dx, demb, B, T, E, V, p, KEEP(p), seed,
                     offsets, slots);
  return check_hip(hipGetLastError(), "embed_tanh_bwd(binned) launch");
}

int vqa_lstm_cell_fwd(const float* xg, const float* hg, const float* c_in, const float* h_in, const int64_t* q_len,
                      int t, float* gates, float* c_out, float* h_out, float* c_final, int64_t cf_ld, int B, int H,
                      vqa_stream_t stream) {
  VQA_REQUIRE(xg && hg && c_in && h_in && q_len && gates && c_out && h_out, "vqa_lstm_cell_fwd: null pointer");
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(grid_for((int64_t)B * H, 256)), dim3(256), 0, STREAM, xg, hg, c_in,
                     h_in, q_len, t, gates, c_out, h_out, c_final, cf_ld, B, H);
  return check_hip(hipGetLastError(), "lstm_cell_fwd launch");
}

int vqa_lstm_cell_bwd(const float* gates, const float* c_in, const float* c_out, const int64_t* q_len, int t,
                      float* dh, float* dc, float* dgates, int B, int H, vqa_stream_t stream) {
  VQA_REQUIRE(gates && c_in && c_out && q_len && dh && dc && dgates, "vqa_lstm_cell_bwd: null pointer");
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(grid_for((int64_t)B * H, 256)), dim3(256), 0, STREAM, gates, c_in,
                     c_out, q_len, t, dh, dc, dgates, B, H);
  return check_hip(hipGetLastError(), "lstm_cell_bwd launch");
}

int vqa_att_score_fwd(const void* xs, int xs_is_bf16, const float* wx, int wx_ld, const float* bx, float* score, int B,
                      int P, int mid, int G, float p, uint64_t seed, const float* qcat, vqa_stream_t stream) {
  set_launch_tag(-1);
  ProfScope prof(VQA_K_ATT_SCORE_FWD, (hipStream_t)stream);
  VQA_REQUIRE(xs && wx && bx && score && mid % 4 == 0 && wx_ld % 4 == 0 && wx_ld >= (qcat ? 2 * mid : mid),
              "vqa_att_score_fwd: bad args");
  const int64_t M = (int64_t)B * P;
  if (xs_is_bf16 && !qcat && mid % 8 == 0 && mid <= 1024 && (reinterpret_cast<uintptr_t>(xs) & 15) == 0) {
    DISPATCH_G(G, hipLaunchKernelGGL((att_score_fwd_bf16_kernel<kG>), dim3(grid_for(M, 8)), dim3(256), 0, STREAM,
                                     static_cast<const uint16_t*>(xs), wx, wx_ld, bx, score, M, P, mid, p, KEEP(p), seed));
  } else if (xs_is_bf16) {
    DISPATCH_G(G, hipLaunchKernelGGL((att_score_fwd_kernel<kG, true>), dim3(grid_for(M, 4)), dim3(256), 0, STREAM, xs, wx,
                                     wx_ld, bx, score, M, P, mid, p, KEEP(p), seed, qcat));
  } else if (!qcat && G <= 2 && mid % 256 == 0 && mid <= 1024 && (reinterpret_cast<uintptr_t>(xs) & 15) == 0) {
#define ROWS_LAUNCH(kG, kIT)                                                                                                  \
  hipLaunchKernelGGL((att_score_fwd_rows_kernel<kG, kIT>), dim3(grid_for(M, 4)), dim3(256), 0, STREAM,                        \
                     static_cast<const float*>(xs), wx, wx_ld, bx, score, M, P, p, KEEP(p), seed)
    switch ((G - 1) * 4 + mid / 256 - 1) {
      case 0: ROWS_LAUNCH(1, 1); break;
      case 1: ROWS_LAUNCH(1, 2); break;
      case 2: ROWS_LAUNCH(1, 3); break;
      case 3: ROWS_LAUNCH(1, 4); break;
      case 4: ROWS_LAUNCH(2, 1); break;
      case 5: ROWS_LAUNCH(2, 2); break;
      case 6: ROWS_LAUNCH(2, 3); break;
      default: ROWS_LAUNCH(2, 4); break;
    }
#undef ROWS_LAUNCH
  } else {
    DISPATCH_G(G, hipLaunchKernelGGL((att_score_fwd_kernel<kG, false>), dim3(grid_for(M, 4)), dim3(256), 0, STREAM, xs, wx,
                                     wx_ld, bx, score, M, P, mid, p, KEEP(p), seed, qcat));
  }
  return check_hip(hipGetLastError(), "att_score_fwd launch");
}

int vqa_att_row_splits(int P) {
  int rs = (P + 127) / 128;
  return rs < 1 ? 1 : (rs > 8 ? 8 : rs);
}

int vqa_att_score_bwd(const float* dscore, const float* wx, int wx_ld, void* xs_inout, int xs_is_bf16, float* dwx_part,
                      float* dq_part, int B, int P, int mid, int G, float p, uint64_t seed, int mode,
                      const float* vprime, const float* qp, vqa_stream_t stream) {
  set_launch_tag(-1);
  ProfScope prof(VQA_K_ATT_SCORE_BWD, (hipStream_t)stream);
  VQA_REQUIRE(dscore && wx && xs_inout && dwx_part && dq_part && mid % 4 == 0 && wx_ld % 4 == 0,
              "vqa_att_score_bwd: bad args");
  VQA_REQUIRE(mode >= 0 && mode <= 2 && (mode != 1 || (vprime && qp)) && (mode != 2 || qp) &&
                  wx_ld >= (mode == 2 ? 2 * mid : mid),
              "vqa_att_score_bwd: mode %d needs its operands (vprime/qp) and a matching wx_ld", mode);
  const int RS = vqa_att_row_splits(P);
  if (xs_is_bf16) {
    DISPATCH_G(G, hipLaunchKernelGGL((att_score_bwd_kernel<kG, true>), dim3(B, RS), dim3(256), 0, STREAM, dscore, wx, wx_ld,
                                     xs_inout, dwx_part, dq_part, P, mid, RS, p, KEEP(p), seed, mode, vprime, qp));
  } else {
    DISPATCH_G(G, hipLaunchKernelGGL((att_score_bwd_kernel<kG, false>), dim3(B, RS), dim3(256), 0, STREAM, dscore, wx, wx_ld,
                                     xs_inout, dwx_part, dq_part, P, mid, RS, p, KEEP(p), seed, mode, vprime, qp));
  }
  return check_hip(hipGetLastError(), "att_score_bwd launch");
}

int vqa_att_apply_fwd(const float* score, const float* vn, float* probs, float* out, int64_t out_ld, int B, int P,
                      int C, int G, vqa_stream_t stream) {
  set_launch_tag(-1);
  ProfScope prof(VQA_K_ATT_APPLY_FWD, (hipStream_t)stream);
  VQA_REQUIRE(score && vn && probs && out, "vqa_att_apply_fwd: null pointer");
  const size_t lds = ((size_t)G * P + 16 + 16 * G * 64) * 4;
  VQA_REQUIRE(lds <= 64 * 1024, "vqa_att_apply_fwd: G*P=%d too large for LDS", G * P);
  if (C % 4 == 0) {
    DISPATCH_G(G, hipLaunchKernelGGL(att_apply_fwd_v4_kernel<kG>, dim3(B, (C + 63) / 64), dim3(256), lds, STREAM, score,
                                     vn, probs, out, out_ld, P, C));
    return check_hip(hipGetLastError(), "att_apply_fwd launch");
  }
  DISPATCH_G(G, hipLaunchKernelGGL(att_apply_fwd_kernel<kG>, dim3(B, (C + 63) / 64), dim3(256), lds, STREAM, score, vn,
                                   probs, out, out_ld, P, C));
  return check_hip(hipGetLastError(), "att_apply_fwd launch");
}

int vqa_att_apply_bwd(const float* dout, int64_t dout_ld, const float* probs, const float* vn, float* dscore,
                      float* dvn, float* dscore_rowsum, int B, int P, int C, int G, vqa_stream_t stream) {
  set_launch_tag(-1);
  ProfScope prof(VQA_K_ATT_APPLY_BWD, (hipStream_t)stream);
  VQA_REQUIRE(dout && probs && vn && dscore && C % 4 == 0 && dout_ld % 4 == 0, "vqa_att_apply_bwd: bad args");
  const int64_t M = (int64_t)B * P;
  DISPATCH_G(G, hipLaunchKernelGGL(att_apply_bwd_rows_kernel<kG>, dim3(grid_for(M, 4)), dim3(256), 0, STREAM, dout,
                                   dout_ld, probs, vn, dscore, dvn, M, P, C));
  int rc = check_hip(hipGetLastError(), "att_apply_bwd_rows launch");
  if (rc) return rc;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(B * G), dim3(256), 0, STREAM, probs, dscore, P, dscore_rowsum);
  return check_hip(hipGetLastError(), "softmax_bwd launch");
}

int vqa_softce_fwd_bwd(const float* logits, int64_t ld, const int64_t* a_idx, const int64_t* a_val, int kmax, int B,
                       int A, float inv_batch, float* loss_rows, float* score_rows, float* dlogits, int64_t dld,
                       vqa_stream_t stream) {
  set_launch_tag(-1);
  ProfScope prof(VQA_K_SOFTCE, (hipStream_t)stream);
  VQA_REQUIRE(logits && a_idx && a_val && loss_rows && score_rows && B > 0 && A > 0 && kmax >= 0,
              "vqa_softce_fwd_bwd: bad args");
  hipLaunchKernelGGL(softce_kernel, dim3(B), dim3(256), 0, STREAM, logits, ld, a_idx, a_val, kmax, A, inv_batch,
                     loss_rows, score_rows, dlogits, dld);
  return check_hip(hipGetLastError(), "softce launch");
}

int64_t vqa_colsum_workspace_bytes(int64_t rows, int cols) { return colsum_ws_bytes(rows, cols); }

int vqa_colsum(const float* x, int64_t ld, const uint8_t* mask, int64_t rows, int cols, float* out, int accumulate,
               float* workspace, int64_t workspace_bytes, vqa_stream_t stream) {
  VQA_REQUIRE(x && out && rows > 0 && cols > 0, "vqa_colsum: bad args");
  VQA_REQUIRE(!mask || ld == cols, "vqa_colsum: masked form needs ld == cols");
  return colsum_launch(x, ld, mask, rows, cols, out, accumulate, workspace, workspace_bytes, STREAM);
}

int vqa_sum_bgp(const float* x, float* out, int B, int G, int P, vqa_stream_t stream) {
  VQA_REQUIRE(x && out, "vqa_sum_bgp: null pointer");
  hipLaunchKernelGGL(sum_bgp_kernel, dim3(G), dim3(1024), 0, STREAM, x, out, B, G, P);
  return check_hip(hipGetLastError(), "sum_bgp launch");
}

int vqa_sum_parts(const float* part, float* out, int batch, int parts, int cols, vqa_stream_t stream) {
  VQA_REQUIRE(part && out && batch > 0 && parts > 0 && cols > 0, "vqa_sum_parts: bad args");
  hipLaunchKernelGGL(sum_parts_kernel, dim3((cols + 255) / 256, batch), dim3(256), 0, STREAM, part, out, parts, cols);
  return check_hip(hipGetLastError(), "sum_parts launch");
}

int vqa_relu_drop_bwd(const float* y, const float* dy, float* dx, int64_t n, float p, uint64_t seed,
                      vqa_stream_t stream) {
  VQA_REQUIRE(y && dy && dx, "vqa_relu_drop_bwd: null pointer");
  hipLaunchKernelGGL(relu_drop_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, STREAM, y, dy, dx, n, p, KEEP(p), seed);
  return check_hip(hipGetLastError(), "relu_drop_bwd launch");
}

int vqa_add2d(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy, int64_t rows,
              int cols, vqa_stream_t stream) {
  VQA_REQUIRE(a && y && rows > 0 && cols > 0, "vqa_add2d: bad args");
  hipLaunchKernelGGL(add2d_kernel, dim3(grid_for(rows * cols, 256)), dim3(256), 0, STREAM, a, lda, b, ldb, y, ldy,
                     rows, cols);
  return check_hip(hipGetLastError(), "add2d launch");
}

int vqa_half_to_float(const void* x_f16, float* y, int64_t n, vqa_stream_t stream) {
  VQA_REQUIRE(x_f16 && y && n > 0, "vqa_half_to_float: bad args");
  VQA_REQUIRE(((uintptr_t)x_f16 % 16) == 0 && ((uintptr_t)y % 16) == 0, "vqa_half_to_float: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(half_to_float_kernel, dim3(grid_for(n / 8 + 1, 256)), dim3(256), 0, STREAM,
                     static_cast<const __half*>(x_f16), y, n);
  return check_hip(hipGetLastError(), "half_to_float launch");
}

int vqa_scale_by(float* x, int64_t n, const float* scalar, vqa_stream_t stream) {
  VQA_REQUIRE(x && scalar && n > 0, "vqa_scale_by: bad args");
  hipLaunchKernelGGL(scale_by_kernel, dim3(grid_for(n, 256)), dim3(256), 0, STREAM, x, n, scalar);
  return check_hip(hipGetLastError(), "scale_by launch");
}

int vqa_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
             float beta2, float eps, int step, float grad_scale, vqa_stream_t stream) {
  set_launch_tag(-1);
  ProfScope prof(VQA_K_ADAM, (hipStream_t)stream);
  VQA_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "vqa_adam: bad args");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256)), dim3(256), 0, STREAM, param, grad, exp_avg, exp_avg_sq, n,
                     step_size, beta1, beta2, eps, inv_sqrt_bc2, grad_scale);
  return check_hip(hipGetLastError(), "adam launch");
}

}  // extern "C"
