// Probe: MFMA rate when every MFMA takes FRESH random bf16 operands read from LDS (no global traffic in the loop) — the
// ceiling a GEMM main loop can reach on random data (clock under power), for the two wave-tile shapes:
//   8 waves x (128x64): 12 ds_read_b128 per 16 MFMAs, two waves per SIMD, free-running or ping-pong barriers
//   4 waves x (128x128): 16 ds_read_b128 per 32 MFMAs, one wave per SIMD, reads slotted between the MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void fill(char *smem, const char *src, int tid, int nthr) {
  for (int i = tid * 16; i < 131072; i += nthr * 16) *reinterpret_cast<uint4 *>(smem + i) = *reinterpret_cast<const uint4 *>(src + i);
  __syncthreads();
}

__global__ __launch_bounds__(512, 2) void k8(const char *src, float *out, int iters, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  fill(smem, src, tid, 512);
  f32x16 c[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) c[i][j][e] = 0.f;
  const bool late = wid >= 4;
  if ((mode & 1) && late) __builtin_amdgcn_s_barrier();
  for (int it = 0; it < iters; ++it) {
    const char *base = smem + (it & 3) * 32768 + (wid & 1) * 8192;
    bf16x8 a[2][4], b[2][2];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const int kc = k2 * 2 + hi;
#pragma unroll
      for (int j = 0; j < 2; ++j) { const int row = j * 32 + l31; b[k2][j] = *reinterpret_cast<const bf16x8 *>(base + 16384 + row * 64 + ((kc ^ ((row >> 2) & 3)) << 4)); }
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int row = i * 32 + l31; a[k2][i] = *reinterpret_cast<const bf16x8 *>(base + row * 64 + ((kc ^ ((row >> 2) & 3)) << 4)); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (mode & 1) { SB(); __builtin_amdgcn_s_barrier(); SB(); }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[k2][j], a[k2][i], c[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if (mode & 1) { SB(); __builtin_amdgcn_s_barrier(); SB(); }
  }
  if ((mode & 1) && !late) __builtin_amdgcn_s_barrier();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += c[i][j][5];
  out[blockIdx.x * 512 + tid] = s;
}

// round 5: the same 8-wave ping-pong loop (same LDS bytes, same flops, same 128 accumulator registers) on v_mfma_f32_16x16x32_bf16:
// 32 MFMAs of 16 cycles per phase instead of 16 of 32 cycles — is the power-limited rate different?
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int ORD>
__global__ __launch_bounds__(512, 2) void k8b(const char *src, float *out, int iters, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;
  fill(smem, src, tid, 512);
  f32x4 c[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) c[i][j][e] = 0.f;
  const bool late = wid >= 4;
  if ((mode & 1) && late) __builtin_amdgcn_s_barrier();
  for (int it = 0; it < iters; ++it) {
    const char *base = smem + (it & 3) * 32768 + (wid & 1) * 8192;
    bf16x8 a[8], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int row = j * 16 + l15; b[j] = *reinterpret_cast<const bf16x8 *>(base + 16384 + row * 64 + ((g4 ^ ((row >> 2) & 3)) << 4)); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int row = i * 16 + l15; a[i] = *reinterpret_cast<const bf16x8 *>(base + row * 64 + ((g4 ^ ((row >> 2) & 3)) << 4)); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (mode & 1) { SB(); __builtin_amdgcn_s_barrier(); SB(); }
    __builtin_amdgcn_s_setprio(1);
    if constexpr (ORD == 1) {  // W fragment stationary across 8 consecutive MFMAs
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], c[i][j], 0, 0, 0);
    } else if constexpr (ORD == 2) {  // operands swapped (A first)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], c[i][j], 0, 0, 0);
    } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], c[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (mode & 1) { SB(); __builtin_amdgcn_s_barrier(); SB(); }
  }
  if ((mode & 1) && !late) __builtin_amdgcn_s_barrier();
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += c[i][j][1];
  out[blockIdx.x * 512 + tid] = s;
}

__global__ __launch_bounds__(256, 1) void k4(const char *src, float *out, int iters, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  fill(smem, src, tid, 256);
  f32x16 c[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) c[i][j][e] = 0.f;
  bf16x8 f[2][8];
  for (int s = 0; s < 2; ++s) for (int q = 0; q < 8; ++q) f[s][q] = *reinterpret_cast<const bf16x8 *>(smem + (q * 32 + l31) * 128 + hi * 16);
  const int xo = (l31 >> 1) & 7;
  for (int it = 0; it < iters; ++it) {
    const char *base = smem + (it & 1) * 65536 + (wid & 1) * 16384 + l31 * 128;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const int cur = sub & 1, nxt = cur ^ 1;
      const int co = ((((sub + 1) & 3) * 2 + hi) ^ xo) << 4;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = q / 4, j = q % 4;
        c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[cur][4 + j], f[cur][i], c[i][j], 0, 0, 0);
        SB();
        if (q < 8) { f[nxt][q] = *reinterpret_cast<const bf16x8 *>(base + (q < 4 ? q * 4096 : 32768 + (q - 4) * 4096) + co); SB(); }
      }
    }
    if (mode & 1) { SB(); __builtin_amdgcn_s_barrier(); SB(); }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += c[i][j][5];
  out[blockIdx.x * 256 + tid] = s;
}

int main() {
  char *d; float *o;
  (void)hipMalloc(&d, 131072); (void)hipMalloc(&o, 256 * 512 * 4);
  static unsigned short h[65536];
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k8), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k8b<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int dist = 0; dist < 3; ++dist) {
    for (int i = 0; i < 65536; ++i) {
      float f;
      if (dist == 0) f = 0.f;
      else if (dist == 1) f = ((rand() % 2001) - 1000) / 1000.f;
      else { float u = 0; for (int q = 0; q < 12; ++q) u += (rand() % 10001) / 10000.f; f = u - 6.f; }  // ~N(0,1)
      unsigned u32; memcpy(&u32, &f, 4); h[i] = (unsigned short)(u32 >> 16);
    }
    (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const char *dn = dist == 0 ? "zeros" : (dist == 1 ? "uniform[-1,1]" : "normal(0,1)");
    for (int mode = 0; mode < 2; ++mode) {
      for (int which = 0; which < 3; ++which) {
        const int iters = 20000, blocks = 256;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        if (which == 0) hipLaunchKernelGGL(k8, dim3(blocks), dim3(512), 131072, 0, d, o, 200, mode);
        else if (which == 2) hipLaunchKernelGGL(k8b<0>, dim3(blocks), dim3(512), 131072, 0, d, o, 200, mode);
        else hipLaunchKernelGGL(k4, dim3(blocks), dim3(256), 131072, 0, d, o, 100, mode);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k8, dim3(blocks), dim3(512), 131072, 0, d, o, iters, mode);
        else if (which == 2) hipLaunchKernelGGL(k8b<0>, dim3(blocks), dim3(512), 131072, 0, d, o, iters, mode);
        else hipLaunchKernelGGL(k4, dim3(blocks), dim3(256), 131072, 0, d, o, iters / 2, mode);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double fl = which != 1 ? (double)blocks * 8 * iters * 16 * 32768.0 : (double)blocks * 4 * (iters / 2) * 64 * 32768.0;
        printf("%-14s %s %s: %.0f TFLOP/s\n", dn, which == 0 ? "8 waves x 128x64 (32x32x16)" : which == 2 ? "8 waves x 128x64 (16x16x32)" : "4 waves x 128x128 (32x32x16)", mode ? (which != 1 ? "ping-pong barriers" : "barrier per K-step ") : "free-running      ", fl / ms / 1e9);
      }
    }
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k8b<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k8b<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int ord = 0; ord < 3; ++ord) {
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto go = [&](int it) {
      if (ord == 0) hipLaunchKernelGGL(k8b<0>, dim3(blocks), dim3(512), 131072, 0, d, o, it, 1);
      else if (ord == 1) hipLaunchKernelGGL(k8b<1>, dim3(blocks), dim3(512), 131072, 0, d, o, it, 1);
      else hipLaunchKernelGGL(k8b<2>, dim3(blocks), dim3(512), 131072, 0, d, o, it, 1);
    };
    go(200);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    go(iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("normal(0,1) 16x16x32 ping-pong, MFMA order %d (0: A-fragment outer, 1: W-fragment outer, 2: operands swapped): %.0f TFLOP/s\n", ord,
           (double)blocks * 8 * iters * 16 * 32768.0 / ms / 1e9);
  }
  return 0;
}
