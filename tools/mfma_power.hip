// Diagnostic micro-benchmark (not part of the library): does the fp32 MFMA rate on this device depend on the
// operand DATA (power management)?  Register-only MFMA loop, 4 waves x 2 workgroups per CU, ~40 ms per run.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline float rnd(unsigned x) {   // uniform in [-2, 2), full mantissa entropy
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return (float)(int)x * (1.0f / 1073741824.0f);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  const unsigned t = blockIdx.x * 256 + threadIdx.x;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    if (MODE == 0) { a[i] = 0.f; b[i] = 0.f; }
    if (MODE == 1) { a[i] = 1.0f; b[i] = 0.5f; }
    if (MODE == 2) { a[i] = rnd(t * 16 + i); b[i] = rnd(t * 16 + 8 + i); }
    if (MODE == 3) { a[i] = (i & 1) ? 0.f : rnd(t * 16 + i); b[i] = rnd(t * 16 + 8 + i); }   // half zeros (post-ReLU like)
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[(s + 1) & 7], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(s + 1) & 7], b[s], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(s + 3) & 7], b[(s + 5) & 7], acc[3], 0, 0, 0);
    }
    if (MODE >= 2 && (it & 63) == 63)     // keep the accumulators bounded without changing the instruction mix much
      for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] *= 0.001f;
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[t] = s;
}

template <int MODE>
void run(const char* name, int iters) {
  const int blocks = 512;
  float* out;
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 32 * 4096.0;
    printf("%-28s run %d  %8.3f ms  %7.1f TF/s  (%.1f%% of 157.3)\n", name, rep, ms, flops / ms / 1e9,
           100 * flops / ms / 1e9 / 157.3);
  }
  hipFree(out);
}

int main() {
  const int iters = 40000;   // 40000 x 32 MFMAs x 64 cycles x 2 waves/SIMD = 164 M cycles ~ 68 ms at 2.4 GHz
  run<0>("zeros", iters);
  run<1>("constants", iters);
  run<2>("random", iters);
  run<3>("random, A half zeros", iters);
  run<0>("zeros again", iters);
  return 0;
}
