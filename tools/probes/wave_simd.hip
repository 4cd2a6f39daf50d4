// Probe: which SIMD does wave w of a 512-thread workgroup land on?  (HW_REG_HW_ID: simd_id = bits 5:4)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out) {
  unsigned v = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = v;
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 64 * 16 * 4); (void)hipMemset(d, 0, 64 * 16 * 4);
  for (int nt : {512, 576, 256}) {
    hipLaunchKernelGGL(probe, dim3(8), dim3(nt), 0, 0, d);
    unsigned h[64 * 16]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("block size %d\n", nt);
    for (int b = 0; b < 8; ++b) {
      printf("  wg %d: simd of waves:", b);
      for (int w = 0; w < nt / 64; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3);
      printf("   (cu %u se %u)\n", (h[b * 16] >> 8) & 15, (h[b * 16] >> 13) & 7);
    }
  }
  return 0;
}
