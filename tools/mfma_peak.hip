// Diagnostic micro-benchmark (not part of the library): what does the fp32 MFMA pipe sustain on this
// device under the access patterns of the GEMM engine?  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float4* __restrict__ g, size_t gmask) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a0 = lane * 0.001f, a1 = 1.f - a0, b0 = 0.5f + a0, b1 = 0.25f;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 1e-4f;
  __syncthreads();
  float4 st[8];
  for (int i = 0; i < 8; ++i) st[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  size_t goff = ((size_t)blockIdx.x * 977 + threadIdx.x) * 8;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 5) {   // LDS-DMA: 8 x (64 lanes x 16 B) per wave per K-step straight into LDS, no VGPRs
      const int wv = threadIdx.x >> 6;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + ((goff + i * 64) & gmask)),
                                         (__attribute__((address_space(3))) void*)(lds + 8192 + (wv * 8 + i) * 256), 16, 0, 0);
      goff += 2048 * 8 + 8;
    }
    if (MODE >= 3 && MODE != 5) {   // staging writes of the previous tile: 32 ds_write_b32 per thread per K-step
      float* w = lds + ((it & 1) ? 0 : 0) + (threadIdx.x >> 3) + 4 * (threadIdx.x & 7) * 129;
#pragma unroll
      for (int i = 0; i < 8; ++i) { float* d = w + 32 * (i & 3) + (i >> 2) * 4224; d[0] = st[i].x; d[129] = st[i].y; d[258] = st[i].z; d[387] = st[i].w; }
    }
    if (MODE == 4) {   // global loads of the next tile: 8 x 16 B per thread per K-step (streaming)
#pragma unroll
      for (int i = 0; i < 8; ++i) st[i] = g[(goff + i * 64) & gmask];
      goff += 2048 * 8 + 8;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (MODE >= 1) {  // fragment reads from LDS, like the engine
        a0 = lds[(s * 2 * 129 + lane) & 8191]; a1 = lds[(s * 2 * 129 + 32 + lane) & 8191];
        b0 = lds[(4224 + s * 2 * 129 + lane) & 8191]; b1 = lds[(4224 + s * 2 * 129 + 32 + lane) & 8191];
      }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
    }
    if (MODE == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE >= 2) __syncthreads();   // one barrier per 64 MFMAs, like one K-step
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks_per_cu, size_t lds_bytes, size_t foot = (size_t)1 << 26) {
  const int iters = 2000;
  const int blocks = 256 * blocks_per_cu;
  float* out;
  hipMalloc(&out, blocks * 256 * 4);
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  static float4* g = nullptr; const size_t gn = (size_t)1 << 26;  // 1 GiB of float4
  if (!g) { hipMalloc(&g, gn * 16); hipMemset(g, 0, gn * 16); }
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds_bytes, 0, out, 10, g, foot - 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds_bytes, 0, out, iters, g, foot - 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 64 * 4096.0;
  printf("%-44s blocks/CU %d  %8.3f ms  %7.1f TF/s  (%.1f%% of 157.3)\n", name, blocks_per_cu, ms, flops / ms / 1e9,
         100 * flops / ms / 1e9 / 157.3);
  hipFree(out);
}

int main() {
  run<0>("pure MFMA, operands in registers", 1, 32768);
  run<0>("pure MFMA, operands in registers", 2, 32768);
  run<1>("MFMA + 4 ds_read_b32 per 4 MFMA", 1, 66048);
  run<1>("MFMA + 4 ds_read_b32 per 4 MFMA", 2, 66048);
  run<2>("MFMA + ds_read + barrier per 64 MFMA", 1, 66048);
  run<2>("MFMA + ds_read + barrier per 64 MFMA", 2, 66048);
  run<3>("  + 32 ds_write_b32 per K-step", 1, 66048);
  run<3>("  + 32 ds_write_b32 per K-step", 2, 66048);
  run<4>("  + 8 global_load_dwordx4/K-step, 1 GiB", 2, 66048);
  run<4>("  + 8 global_load_dwordx4/K-step, 64 MiB", 2, 66048, (size_t)1 << 22);
  run<4>("  + 8 global_load_dwordx4/K-step, 2 MiB", 1, 66048, (size_t)1 << 17);
  run<4>("  + 8 global_load_dwordx4/K-step, 2 MiB", 2, 66048, (size_t)1 << 17);
  run<5>("  LDS-DMA 8 x dwordx4/K-step, 2 MiB", 1, 66048, (size_t)1 << 17);
  run<5>("  LDS-DMA 8 x dwordx4/K-step, 2 MiB", 2, 66048, (size_t)1 << 17);
  run<5>("  LDS-DMA 8 x dwordx4/K-step, 1 GiB", 2, 66048);
  return 0;
}
