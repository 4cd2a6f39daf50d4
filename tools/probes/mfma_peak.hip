// Probe: sustained MFMA rate (32x32x16 bf16) with register-resident operands, no memory traffic:
// the practical ceiling (clock under load) for operand data of different statistics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512, 2) void k(const bf16x8* in, float* out, int iters, int waves_active) {
  const int wid = threadIdx.x >> 6;
  bf16x8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = in[(threadIdx.x * 7 + i * 3 + blockIdx.x) & 4095];
  for (int i = 0; i < 2; ++i) b[i] = in[(threadIdx.x * 5 + i * 11 + blockIdx.x * 3) & 4095];
  f32x16 acc[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (wid < waves_active) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
  }
  float s = 0; for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  bf16x8* d; float* o; (void)hipMalloc(&d, 4096 * 16); (void)hipMalloc(&o, 256 * 8 * 512 * 4);
  unsigned short h[4096 * 8];
  for (int mode = 0; mode < 3; ++mode) {
    for (int i = 0; i < 4096 * 8; ++i) {
      float f = mode == 0 ? 0.f : (mode == 1 ? ((rand() % 2001) - 1000) / 1000.f : ((rand() % 2001) - 1000) / 50000.f);
      unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16);
    }
    (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int wa : {8, 4}) {
      const int iters = 20000, blocks = 256;
      hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, o, 100, wa);
      (void)hipDeviceSynchronize();
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, o, iters, wa);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      double fl = (double)blocks * wa * iters * 8 * 32768.0;
      printf("data %s waves/WG %d: %.2f ms  %.0f TFLOP/s\n", mode == 0 ? "zeros" : (mode == 1 ? "uniform[-1,1]" : "uniform[-0.02,0.02]"), wa, ms, fl / ms / 1e9);
    }
  }
  return 0;
}
