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
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char *dst, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)dst, 16, voff, soff, 0, 0);
}
#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void fill(char *smem, const char *src, int tid, int nthr) {
  for (int i = tid * 16; i < 131072; i += nthr * 16) *reinterpret_cast<uint4 *>(smem + i) = *reinterpret_cast<const uint4 *>(src + i);
  __syncthreads();
}

template <int MODE>
__device__ __forceinline__ void k8_body(const char *src, float *out, int iters, const char *big) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  fill(smem, src, tid, 512);
  f32x16 c[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) c[i][j][e] = 0.f;
  const bool late = wid >= 4;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)big, 0, 0x7fffffff, 0x00020000);
  unsigned pv[8];
  for (int i = 0; i < 8; ++i) {
    const int row = (blockIdx.x & 7) * 512 + (wid * 8 + i) * 8 + (lane >> 3);
    pv[i] = (MODE & 8) ? (unsigned)(lane * 16) : (unsigned)row * 2816u + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
  }
  char *dmadst = smem + 131072 + wid * 4096;
  bf16x8 a[2][4], b[2][2];
  auto reads = [&](int it) {
    const char *base = smem + (it & 3) * 32768 + (wid & 1) * 8192;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const int kc = k2 * 2 + hi;
#pragma unroll
      for (int j = 0; j < 2; ++j) { const int row = j * 32 + l31; b[k2][j] = *reinterpret_cast<const bf16x8 *>(base + 16384 + row * 64 + ((kc ^ ((row >> 2) & 3)) << 4)); }
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int row = i * 32 + l31; a[k2][i] = *reinterpret_cast<const bf16x8 *>(base + row * 64 + ((kc ^ ((row >> 2) & 3)) << 4)); }
    }
  };
  auto dma = [&](int st) {
#pragma unroll
    for (int i = 0; i < 8; ++i) dma16(r, dmadst + (i & 3) * 1024, pv[i], (MODE & 8) ? 0 : (st % 22) * 128);
  };
  auto mma = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[k2][j], a[k2][i], c[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
#define BAR() do { SB(); __builtin_amdgcn_s_barrier(); SB(); } while (0)
  if (late) BAR();
  for (int st = 0; st < iters; ++st) {
    reads(2 * st);
    if (MODE & 2) dma(st);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BAR();
    if (MODE & 4) dma(st);
    mma();
    BAR();
    reads(2 * st + 1);
    if (MODE & 6) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BAR();
    mma();
    BAR();
  }
  if (!late) BAR();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += c[i][j][5];
  out[blockIdx.x * 512 + tid] = s;
}
__global__ __launch_bounds__(512, 2) void k8_0(const char *src, float *out, int iters, const char *big) { k8_body<0>(src, out, iters, big); }
__global__ __launch_bounds__(512, 2) void k8_2(const char *src, float *out, int iters, const char *big) { k8_body<2>(src, out, iters, big); }
__global__ __launch_bounds__(512, 2) void k8_4(const char *src, float *out, int iters, const char *big) { k8_body<4>(src, out, iters, big); }
__global__ __launch_bounds__(512, 2) void k8_10(const char *src, float *out, int iters, const char *big) { k8_body<10>(src, out, iters, big); }

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
  char *d; float *o; char *big;
  (void)hipMalloc(&big, 8 * 512 * 2816 + 65536); (void)hipMemset(big, 0x3c, 8 * 512 * 2816 + 65536);
  (void)hipMalloc(&d, 131072); (void)hipMalloc(&o, 256 * 512 * 4);
  static unsigned short h[65536];
  for (int i = 0; i < 65536; ++i) {
    float u = 0; for (int q = 0; q < 12; ++q) u += (rand() % 10001) / 10000.f; float f = u - 6.f;
    unsigned u32; memcpy(&u32, &f, 4); h[i] = (unsigned short)(u32 >> 16);
  }
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  typedef void (*kern_t)(const char *, float *, int, const char *);
  kern_t ks[4] = {k8_0, k8_2, k8_4, k8_10};
  const char *names[4] = {"no DMA", "DMA issued in read phase (8 rows x 128 B, L2 hits)", "DMA issued in MFMA phase", "DMA in read phase, one hot 1-KiB line group (L1 hits)"};
  for (int rep = 0; rep < 2; ++rep)
  for (int v = 0; v < 4; ++v) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ks[v]), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    const int iters = 10000, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(ks[v], dim3(blocks), dim3(512), 163840, 0, d, o, 100, big);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(ks[v], dim3(blocks), dim3(512), 163840, 0, d, o, iters, big);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 8 * iters * 32 * 32768.0;
    printf("ping-pong 8 waves x 128x64, normal(0,1) operands, %s: %.0f TFLOP/s\n", names[v], fl / ms / 1e9);
  }
  return 0;
}
