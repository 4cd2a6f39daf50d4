// Probe: upper bound of a ONE-wave-per-SIMD GEMM main loop (4 waves per CU, software-pipelined in one instruction stream):
// per K sub-step of 16 a wave issues TM*TN MFMAs (32x32x16), TM+TN fragment reads for the NEXT sub-step and its share of the
// LDS-DMA (8 rows x 128 B pieces), interleaved one non-MFMA instruction per MFMA.  Addresses mimic the real kernel; results are
// meaningless.  TN = 2: CU tile 256x128 (48 KiB per K-step of 64, 3 stages); TN = 4: CU tile 256x256 (64 KiB, 2 stages).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define SB() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char *dst, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)dst, 16, voff, soff, 0, 0);
}

template <int TN>
__global__ __launch_bounds__(256, 1) void k(const char *src, float *out, int iters, int mode, int ld_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TM = 4, NF = TM + TN;
  constexpr int STEP = (256 + 64 * TN) * 128, NST = TN == 2 ? 3 : 2;
  constexpr int PPS = (256 + 64 * TN) / 8 / 4;  // pieces per wave per K-step (12 or 16)
  constexpr int PPSUB = PPS / 4;                // per sub-step (3 or 4)
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7fffffff, 0x00020000);
  unsigned pa[PPS];
  for (int i = 0; i < PPS; ++i) {
    const int row = (blockIdx.x % 64) * 512 + (wid * PPS + i) * 8 + (lane >> 3);
    pa[i] = (unsigned)row * (unsigned)ld_bytes + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
  }
  f32x16 c[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) c[i][j][e] = 0.f;
  i32x4 f[2][NF];
  for (int s = 0; s < 2; ++s) for (int q = 0; q < NF; ++q) f[s][q] = (i32x4){lane, q, s, 1};
  const unsigned xo = (unsigned)((l31 >> 1) & 7);
  int stage = 0;
  for (int it = 0; it < iters; ++it) {
    const int n = it % 22;
    const unsigned rbase = (unsigned)(stage * STEP) + (unsigned)(l31 * 128);
    const int dstage = stage + (NST - 1) >= NST ? stage + (NST - 1) - NST : stage + (NST - 1);
    char *dst = smem + dstage * STEP + wid * PPS * 1024;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const int cur = sub & 1, nxt = cur ^ 1;
      const unsigned kc = (unsigned)((((sub + 1) & 3) * 2 + hi) ^ xo) << 4;
#pragma unroll
      for (int q = 0; q < TM * TN; ++q) {
        const int i = q / TN, j = q % TN;
        c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[cur][TM + j]), __builtin_bit_cast(bf16x8, f[cur][i]), c[i][j], 0, 0, 0);
        SB();
        if (q < NF && (mode & 1)) {
          const unsigned a = rbase + (unsigned)(q * 4096) + kc;
          asm volatile("ds_read_b128 %0, %1" : "=v"(f[nxt][q]) : "v"(a));
          SB();
        }
        if (q >= TM * TN - PPSUB && (mode & 2)) {
          const int pi = sub * PPSUB + (q - (TM * TN - PPSUB));
          dma16(r, dst + pi * 1024, pa[pi], n * 128);
          SB();
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SB();
    }
    if (mode & 2) {
      if (PPS == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
    if (mode & 4) { SB(); __builtin_amdgcn_s_barrier(); SB(); }
    stage = stage + 1 == NST ? 0 : stage + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) s += c[i][j][5];
  out[blockIdx.x * 256 + tid] = s + (float)(f[0][0][0] + f[1][1][1]);
}

template <int TN>
void run(const char *d, float *o) {
  const int smem = TN == 2 ? 3 * 49152 : 2 * 65536;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<TN>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  for (int mode : {0, 1, 2, 3, 7}) {
    const int iters = 10000, blocks = 256;
    hipLaunchKernelGGL(k<TN>, dim3(blocks), dim3(256), smem, 0, d, o, 200, mode, 2816);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<TN>, dim3(blocks), dim3(256), smem, 0, d, o, iters, mode, 2816);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns_it = ms * 1e6 / iters, fl = 4.0 * 4 * 4 * TN * 32768.0;
    printf("wave tile 128x%d mode %d [%s%s%s]: %.1f ns per K-step per CU, mfma %.0f TF/s (chip), dma %.0f B/ns/CU\n", 32 * TN, mode, (mode & 1) ? "read " : "",
           (mode & 2) ? "dma " : "", (mode & 4) ? "barrier" : "", ns_it, fl * 256 / ns_it / 1e3, (mode & 2) ? (256 + 64 * TN) * 128.0 / ns_it : 0.0);
  }
}

int main() {
  char *d; float *o;
  const size_t bytes = (size_t)64 * 512 * 2816 + 65536;
  (void)hipMalloc(&d, bytes); (void)hipMemset(d, 1, bytes); (void)hipMalloc(&o, 256 * 256 * 4);
  run<2>(d, o);
  run<4>(d, o);
  return 0;
}
