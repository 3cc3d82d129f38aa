// Diagnostic micro-benchmark (not part of the library): the patch convolution's inner loop -- fragments read from LDS with
// ds_read_b128, 128 accumulator VGPRs per wave, 8 waves per CU -- on the two bf16 MFMA shapes.  Per unit of 32 k (two 16-channel
// slices of one tap): 32x32x16: 2 x (4 A + 2 B reads, 8 MFMAs of 32 cycles); 16x16x32: 8 A + 4 B reads, 32 MFMAs of 16 cycles.
// Same bytes from LDS, same FLOPs.  Data: A ~ post-ReLU activations (half zeros), B ~ weights, both bf16.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, int zeros) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];      // 64 KiB of operand data
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 65536 / 4; i += 512) {
    const uint32_t h = mix(i * 2654435761u + blockIdx.x);
    // two bf16 per dword: values in [-2, 2) with 8 random mantissa bits; A half (first 32 KiB): half of them zero
    uint32_t lo = 0x3f800000u ^ ((h & 0xffu) << 15) ^ ((h >> 8 & 1u) << 31), hi = 0x3f800000u ^ ((h >> 9 & 0xffu) << 15) ^ ((h >> 17 & 1u) << 31);
    if (zeros && i < 8192) { if (h >> 20 & 1) lo = 0; if (h >> 21 & 1) hi = 0; }
    reinterpret_cast<uint32_t*>(smem)[i] = (lo >> 16) | (hi & 0xffff0000u);
  }
  __syncthreads();
  const char* pa = smem + (wave & 3) * 4096 + lane * 16;
  const char* pb = smem + 32768 + (wave >> 2) * 4096 + lane * 16;
  float s = 0.f;
  if (SHAPE == 0) {
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {                      // 4 k-steps of 16
        bf16x8 a[4], b[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(pa + ((t * 4 + i) & 15) * 1024 + (it & 1) * 16384);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(pb + ((t * 2 + j) & 3) * 1024 + (it & 1) * 16384);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if ((it & 15) == 15) for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] *= 0.0625f;
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  } else {
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {                      // 2 k-steps of 32
        bf16x8 a[8], b[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const bf16x8*>(pa + ((t * 8 + i) & 15) * 1024 + (it & 1) * 16384);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(pb + ((t * 4 + j) & 3) * 1024 + (it & 1) * 16384);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if ((it & 15) == 15) for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] *= 0.0625f;
    }
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int SHAPE>
static void run(const char* name, int zeros) {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 40000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(512), 65536, 0, out, iters, zeros);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double flops = 256.0 * 8 * iters * 32.0 * 32768.0;      // per iteration and wave: 32 MFMAs x 32x32x16 (or 128 x 16x16x32)
  printf("%-28s zeros=%d  %8.2f ms  %8.1f TFLOP/s\n", name, zeros, best, flops / best / 1e9);
  hipFree(out);
}

int main() {
  for (int z = 0; z < 2; ++z) {
    run<0>("v_mfma_f32_32x32x16_bf16", z);
    run<1>("v_mfma_f32_16x16x32_bf16", z);
  }
  return 0;
}
