// Diagnostic: wave-specialised K-step skeleton (4 MFMA waves + 4 loader waves, one barrier per K-step).
// Variants switch off parts of the loader to see what the MFMA waves end up waiting for.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LOADS: 0 none, 1 L2-resident, 2 streaming.  WRITES: ds_write on/off.  NLOAD loads per thread per step.
template <int LOADS, int WRITES, int NLOAD, int BIG = 0>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, const float4* __restrict__ g, size_t gmask) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16512; i += 512) lds[i] = i * 1e-4f;
  __syncthreads();
  if (threadIdx.x >= 256) {
    if (LOADS == 4) return;                      // variant: the extra waves leave at once
    const int t = threadIdx.x - 256;
    float4 st[NLOAD];
    for (int i = 0; i < NLOAD; ++i) st[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    size_t goff = ((size_t)blockIdx.x * 977 + (t >> 3)) * 16 + (t & 7);   // 8 lanes share a 128-B line, rows 256 B apart
    for (int it = 0; it < iters; ++it) {
      float* w = lds + ((it & 1) ? (BIG ? 12384 : 8256) : 0) + (t >> 3) + 4 * (t & 7) * 129;
      if (WRITES) {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) { float* d = w + 32 * (i & 3) + (i >> 2) * 4128; d[0] = st[i].x; d[129] = st[i].y; d[258] = st[i].z; d[387] = st[i].w; }
      }
      if (LOADS == 3) {   // LDS-DMA: straight into the other buffer, no VGPRs, no ds_write
        const int wv = (threadIdx.x >> 6) - 4;
        float* dst = lds + ((it & 1) ? 8256 : 0);
#pragma unroll
        for (int i = 0; i < NLOAD; ++i)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + ((goff + i * 32 * 16) & gmask)),
                                           (__attribute__((address_space(3))) void*)(dst + (wv * NLOAD + i) * 256), 16, 0, 0);
        goff += 61 * 16;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (LOADS) {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) st[i] = g[(goff + i * 32 * 16) & gmask];
        goff += 61 * 16;
      }
      __syncthreads();
    }
    if (st[0].x == 12345.f) out[0] = 1.f;
    return;
  }
  if (BIG) {
    f32x16 acc8[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc8[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const float* base = lds + ((it & 1) ? 0 : 12384);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        float a[4], b[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = base[s * 2 * 257 + 32 * i + lane];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = base[8224 + s * 2 * 129 + 32 * j + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc8[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc8[i * 2 + j], 0, 0, 0);
      }
      __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc8[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    return;
  }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    const float* base = lds + ((it & 1) ? 0 : 8256);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float a0 = base[s * 2 * 129 + lane], a1 = base[s * 2 * 129 + 32 + lane];
      const float b0 = base[4128 + s * 2 * 129 + lane], b1 = base[4128 + s * 2 * 129 + 32 + lane];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float4* g = nullptr;
template <int LOADS, int WRITES, int NLOAD, int BIG = 0>
void run(const char* name, int blocks_per_cu, size_t foot) {
  const int iters = 2000, blocks = 256 * blocks_per_cu;
  const size_t lds_bytes = BIG ? 99072 : 66048;
  float* out;
  hipMalloc(&out, blocks * 256 * 4);
  if (!g) { hipMalloc(&g, ((size_t)1 << 26) * 16); hipMemset(g, 0, ((size_t)1 << 26) * 16); }
  auto kern = k<LOADS, WRITES, NLOAD, BIG>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds_bytes, 0, out, 10, g, foot - 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds_bytes, 0, out, iters, g, foot - 1);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * (BIG ? 128 : 64) * 4096.0;
  printf("%-52s blocks/CU %d %8.3f ms %7.1f TF/s (%.1f%%)\n", name, blocks_per_cu, ms, flops / ms / 1e9, 100 * flops / ms / 1e9 / 157.3);
  hipFree(out);
}

int main() {
  const size_t L2 = (size_t)1 << 17, BIG = (size_t)1 << 26;
  run<4, 0, 8>("WS: extra waves exit immediately", 2, L2);
  run<0, 0, 8>("WS: loaders idle (barrier only)", 1, L2);
  run<0, 0, 8>("WS: loaders idle (barrier only)", 2, L2);
  run<0, 1, 8>("WS: 32 ds_write_b32/thread/step", 2, L2);
  run<1, 0, 8>("WS: 8 loads (2 MiB footprint), no writes", 2, L2);
  run<1, 1, 8>("WS: 8 loads (2 MiB) + writes", 1, L2);
  run<1, 1, 8>("WS: 8 loads (2 MiB) + writes", 2, L2);
  run<1, 1, 8>("WS: 8 loads (1 GiB stream) + writes", 2, BIG);
  run<1, 1, 4>("WS: 4 loads (2 MiB) + 16 writes", 2, L2);
  run<0, 0, 12, 1>("WS 256x128: loaders idle", 1, L2);
  run<1, 1, 12, 1>("WS 256x128: 12 loads (2 MiB) + 48 writes", 1, L2);
  run<1, 1, 12, 1>("WS 256x128: 12 loads (1 GiB) + 48 writes", 1, BIG);
  run<3, 0, 8>("WS: 8 LDS-DMA (2 MiB), wait same step", 1, L2);
  run<3, 0, 8>("WS: 8 LDS-DMA (2 MiB), wait same step", 2, L2);
  run<3, 0, 8>("WS: 8 LDS-DMA (1 GiB), wait same step", 2, BIG);
  return 0;
}
