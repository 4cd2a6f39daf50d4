// Probe: LDS-side cost of the GEMM's two LDS clients — fragment reads (ds_read_b128) and LDS-DMA landing
// (buffer_load_dwordx4 ... lds) — alone and together, in the access pattern of gemm_pp3_kernel (1-KiB pieces of 16 rows x
// 64 B, rows 2816 B apart in memory).  Prints bytes per ns per CU; 2.4 GHz x 256 B/clk = 614 B/ns is the ds_read_b128 peak.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// mode bit 0: 12 ds_read_b128 per iteration; bit 1: 4 LDS-DMA pieces per iteration; bit 2: source advances along K (L2 hits
// instead of L1 hits); bit 3: 16 MFMAs per iteration too
__global__ __launch_bounds__(512, 2) void k(const char *src, float *out, int iters, int mode, int ld_bytes, int pat) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5, prow = lane >> 2, pslot = lane & 3;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7fffffff, 0x00020000);
  unsigned pa[4];
  // pat: bytes of one memory row fetched by one instruction = 64 << pat (64: 16 rows x 4 lanes ... 1024: one row, 64 lanes)
  const int lpr = 4 << pat, rpi = 64 / lpr;  // lanes per row, rows per instruction
  for (int i = 0; i < 4; ++i) {
    const int row = (blockIdx.x % 64) * 512 + (wid * 4 + i) * rpi + lane / lpr;
    pa[i] = (unsigned)row * (unsigned)ld_bytes + (((lane % lpr) ^ ((row >> 2) & 3)) << 4);
  }
  const int kadv = 64 << pat;
  i32x4 acc = {0, 0, 0, 0};
  f32x16 c[8];
  bf16x8 fa, fb;
  for (int q = 0; q < 8; ++q) for (int e = 0; e < 16; ++e) c[q][e] = 0.f;
  for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)(float)(lane + e); fb[e] = (__bf16)(float)(lane * 3 + e); }
  const int nk = 44;
  for (int it = 0; it < iters; ++it) {
    const int n = it % nk;
    const unsigned rbase = (unsigned)((it & 3) * 32768 + (wid & 1) * 8192);
    if (mode & 1) {
      i32x4 v[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const int row = (j % 6) * 32 + l31, kc = (j / 6) * 2 + hi;
        const unsigned a = rbase + row * 64 + ((kc ^ ((row >> 2) & 3)) << 4);
        asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(a));
      }
      if (mode & 2) {
        char *dst = smem + ((it + 2) & 3) * 32768 + wid * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)(dst + i * 1024), 16, pa[i], (mode & 4) ? (n * kadv) % 2816 : 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 12; ++j) acc += v[j];
      if (mode & 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else if (mode & 2) {
      char *dst = smem + ((it + 2) & 3) * 32768 + wid * 4096;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)(dst + i * 1024), 16, pa[i], (mode & 4) ? (n * kadv) % 2816 : 0, 0, 0);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    if (mode & 8) {
#pragma unroll
      for (int q = 0; q < 16; ++q) c[q & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c[q & 7], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0;
  for (int q = 0; q < 8; ++q) s += c[q][5];
  out[blockIdx.x * 512 + tid] = (float)(acc[0] + acc[1] + acc[2] + acc[3]) + s;
}

int main() {
  const int ld = 2816;
  char *d; float *o;
  const size_t bytes = (size_t)64 * 512 * ld + 65536;
  (void)hipMalloc(&d, bytes); (void)hipMemset(d, 1, bytes); (void)hipMalloc(&o, 256 * 512 * 4);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int pat = 0; pat < 5; ++pat)
  for (int mode : {2, 6, 14, 15}) {
    const int iters = 20000, blocks = 256;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 131072, 0, d, o, 200, mode, ld, pat);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 131072, 0, d, o, iters, mode, ld, pat);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns_it = ms * 1e6 / iters;
    const double rd = (mode & 1) ? 8 * 12 * 1024.0 : 0, dma = (mode & 2) ? 8 * 4 * 1024.0 : 0, fl = (mode & 8) ? 8 * 16 * 32768.0 : 0;
    printf("row bytes/instr %4d mode %2d [%s%s%s%s]: %.1f ns/iter/CU  reads %.0f B/ns  dma %.0f B/ns  mfma %.0f TF/s (chip)\n", 64 << pat, mode, (mode & 1) ? "read " : "",
           (mode & 2) ? "dma " : "", (mode & 4) ? "advK " : "", (mode & 8) ? "mfma" : "", ns_it, rd / ns_it, dma / ns_it, fl * 256 / ns_it / 1e3);
  }
  return 0;
}
