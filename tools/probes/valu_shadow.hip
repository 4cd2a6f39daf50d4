// Probe (round 6): can VALU work hide under MFMAs in the 8-wave ping-pong loop of gemm_pp4_kernel?  The loop of mfma_lds_ceiling.hip
// (k8b: 16x16x32, two wave groups alternating read phase / MFMA phase, fed from LDS with N(0,1) operands) with NV plain VALU
// instructions added per phase, in one of three places:
//   place 0: in the READ phase, behind the 12 fragment reads (the partner wave of the SIMD is in its MFMA phase)
//   place 1: in the MFMA phase, NV / 32 behind each of the wave's own 32 MFMAs
//   place 2: at the START of the wave's read phase, in front of its fragment reads (the reads are issued late)
// prio bits: 1 = MFMA phase at s_setprio 1 (the product kernel), 2 = read-phase fillers at s_setprio 2 (above the MFMA phase).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int NV, int PLACE, int PRIO>
__global__ __launch_bounds__(512, 2) void k(const char *src, float *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;
  for (int i = tid * 16; i < 131072; i += 512 * 16) *reinterpret_cast<uint4 *>(smem + i) = *reinterpret_cast<const uint4 *>(src + i);
  __syncthreads();
  f32x4 c[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) c[i][j][e] = 0.f;
  float v[8];
  for (int q = 0; q < 8; ++q) v[q] = 1.0f + q * 0.125f + lane * 1e-3f;
  const float mul = 0.999f;
  const bool late = wid >= 4;
  if (late) __builtin_amdgcn_s_barrier();
  for (int it = 0; it < iters; ++it) {
    const char *base = smem + (it & 3) * 32768 + (wid & 1) * 8192;
    bf16x8 a[8], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int row = j * 16 + l15; b[j] = *reinterpret_cast<const bf16x8 *>(base + 16384 + row * 64 + ((g4 ^ ((row >> 2) & 3)) << 4)); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int row = i * 16 + l15; a[i] = *reinterpret_cast<const bf16x8 *>(base + row * 64 + ((g4 ^ ((row >> 2) & 3)) << 4)); }
    if constexpr (PLACE == 0 && NV > 0) {
      SB();
      if constexpr (PRIO & 2) __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int n = 0; n < NV; ++n) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[n & 7]) : "v"(mul));
      if constexpr (PRIO & 2) __builtin_amdgcn_s_setprio(0);
      SB();
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    SB(); __builtin_amdgcn_s_barrier(); SB();
    if constexpr (PRIO & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], c[i][j], 0, 0, 0);
        if constexpr (PLACE == 1 && NV > 0) {
          constexpr int PER = NV / 32, EXTRA = NV % 32;
          SB();
#pragma unroll
          for (int n = 0; n < PER; ++n) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[(i * 4 + j + n) & 7]) : "v"(mul));
          if ((i * 4 + j) < EXTRA) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[(i * 4 + j) & 7]) : "v"(mul));
          SB();
        }
      }
    if constexpr (PRIO & 1) __builtin_amdgcn_s_setprio(0);
    SB(); __builtin_amdgcn_s_barrier(); SB();
    if constexpr (PLACE == 2 && NV > 0) {  // = the head of the next read phase
#pragma unroll
      for (int n = 0; n < NV; ++n) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[n & 7]) : "v"(mul));
      SB();
    }
  }
  if (!late) __builtin_amdgcn_s_barrier();
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += c[i][j][1];
  for (int q = 0; q < 8; ++q) s += v[q];
  out[blockIdx.x * 512 + tid] = s;
}

static char *d; static float *o;
template <int NV, int PLACE, int PRIO>
void run() {
  auto kk = k<NV, PLACE, PRIO>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kk), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int iters = 20000, blocks = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kk, dim3(blocks), dim3(512), 131072, 0, d, o, 400);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 2; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kk, dim3(blocks), dim3(512), 131072, 0, d, o, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  // per iteration a wave runs one read phase + one MFMA phase (32 MFMAs = 512 pipe cycles)
  printf("place %d prio %d NV %3d per phase: %6.0f TFLOP/s  %7.1f us  (%.1f ns per phase pair)\n", PLACE, PRIO, NV, (double)blocks * 8 * iters * 16 * 32768.0 / best / 1e9,
         best * 1e3, best * 1e6 / iters);
  fflush(stdout);
}

int main(int argc, char **argv) {
  (void)hipMalloc(&d, 131072); (void)hipMalloc(&o, 256 * 512 * 4);
  static unsigned short h[65536];
  const bool zeros = argc > 1 && !strcmp(argv[1], "zeros");  // zero MFMA operands: the matrix pipe draws little power, the clock stays at 2.4 GHz — separates "issue port" from "power"
  for (int i = 0; i < 65536; ++i) {
    float u = 0; for (int q = 0; q < 12; ++q) u += (rand() % 10001) / 10000.f;
    float f = zeros ? 0.f : u - 6.f; unsigned u32; memcpy(&u32, &f, 4); h[i] = (unsigned short)(u32 >> 16);
  }
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  printf("== MFMA operands: %s\n", zeros ? "zeros" : "normal(0,1)");
  if (zeros) {
    run<0, 0, 1>(); run<16, 0, 1>(); run<32, 0, 1>(); run<64, 0, 1>(); run<96, 0, 1>(); run<128, 0, 1>();
    run<0, 0, 0>(); run<32, 0, 0>(); run<64, 0, 0>(); run<128, 0, 0>();
    run<32, 1, 1>(); run<64, 1, 1>(); run<128, 1, 1>();
    return 0;
  }
  run<0, 0, 1>(); run<0, 0, 0>();
  printf("-- read-phase fillers, MFMA phase at prio 1 (product)\n");
  run<16, 0, 1>(); run<32, 0, 1>(); run<48, 0, 1>(); run<64, 0, 1>(); run<96, 0, 1>(); run<128, 0, 1>();
  printf("-- read-phase fillers, no priorities\n");
  run<16, 0, 0>(); run<32, 0, 0>(); run<48, 0, 0>(); run<64, 0, 0>(); run<96, 0, 0>(); run<128, 0, 0>();
  printf("-- read-phase fillers at prio 2, MFMA phase at prio 1\n");
  run<16, 0, 3>(); run<32, 0, 3>(); run<48, 0, 3>(); run<64, 0, 3>(); run<96, 0, 3>(); run<128, 0, 3>();
  printf("-- read-phase fillers at prio 2, MFMA phase at prio 0\n");
  run<32, 0, 2>(); run<64, 0, 2>(); run<96, 0, 2>();
  printf("-- fillers behind the wave's own MFMAs (prio 1)\n");
  run<16, 1, 1>(); run<32, 1, 1>(); run<48, 1, 1>(); run<64, 1, 1>(); run<96, 1, 1>(); run<128, 1, 1>();
  printf("-- fillers behind the wave's own MFMAs (no prio)\n");
  run<32, 1, 0>(); run<64, 1, 0>(); run<96, 1, 0>();
  printf("-- fillers at the head of the read phase, in front of the reads (prio 1 on the MFMA phase)\n");
  run<32, 2, 1>(); run<64, 2, 1>(); run<128, 2, 1>();
  return 0;
}
