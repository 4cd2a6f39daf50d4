// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  Each lane supplies an 8-byte-aligned LDS address;
// prints which (lane, element) of the INPUT each (lane, element) of the OUTPUT received.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 4];
  const int lane = threadIdx.x;
  // lane l owns lds[4l .. 4l+3]; value encodes (lane, element) = lane*4 + e
  for (int e = 0; e < 4; ++e) lds[lane * 4 + e] = (uint16_t)(lane * 4 + e);
  __syncthreads();
  uint32_t addr = (uint32_t)(uintptr_t)(&lds[lane * 4]);  // LDS byte address (low 32 bits of the flat shared ptr)
  addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)(&lds[lane * 4]);
  uint64_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (uint16_t)(v >> (16 * e));
}
int main() {
  uint16_t* d; printf("malloc %d\n", (int)hipMalloc(&d, 512)); hipMemset(d, 0xff, 512);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  printf("launch %d sync %d\n", (int)hipGetLastError(), (int)hipDeviceSynchronize());
  uint16_t h[256]; printf("copy %d\n", (int)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost));
  for (int l = 0; l < 64; ++l) {
    printf("L%02d:", l);
    for (int e = 0; e < 4; ++e) printf(" %2d.%d", h[l * 4 + e] / 4, h[l * 4 + e] % 4);
    printf(l % 4 == 3 ? "\n" : "  |  ");
  }
  return 0;
}
