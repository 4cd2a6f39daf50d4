// Probe: LDS read rate of the K-fragment patterns of the frame-attention kernels on 176-byte rows.
//   0: two ds_read_b64 per k-step (d = 32 ks + 16 hf + 4 g .. + 3): the pattern in attn_frame_kernel
//   1: one ds_read_b128 per k-step (d = 32 ks + 8 g .. + 7)
//   2: ds_read_b128, lane * 16 (linear: the conflict-free reference)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) unsigned u2;
typedef __attribute__((ext_vector_type(4))) unsigned u4;
template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned long long *out, int iters) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
    for (int i = threadIdx.x; i < 48 * 1024 / 4; i += 512) ((unsigned *)smem)[i] = i;
    __syncthreads();
    const int prow = l15 < 8 ? 2 * l15 : 2 * l15 - 15;
    unsigned addr = (unsigned)(uintptr_t)smem;
    if (MODE == 0) addr += prow * 176 + g * 8;
    if (MODE == 1) addr += prow * 176 + g * 16;
    if (MODE == 2) addr += lane * 16;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if constexpr (MODE == 0) {
                u2 x[6];
#pragma unroll
                for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
                        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[ks * 2 + hf]) : "v"(addr), "i"(t * 16 * 176 + ks * 64 + hf * 32));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]));
                for (int k = 0; k < 6; ++k) acc += x[k].x + x[k].y;
            } else {
                u4 x[3];
#pragma unroll
                for (int ks = 0; ks < 3; ++ks)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[ks]) : "v"(addr), "i"((MODE == 1 ? t * 16 * 176 : t * 1024) + ks * 64));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));
                for (int k = 0; k < 3; ++k) acc += x[k].x + x[k].y + x[k].z + x[k].w;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 0x12345678u) out[1000] = acc;
}
template <int MODE>
void run(const char *name, unsigned long long *d, int waves) {
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipLaunchKernelGGL((probe<MODE>), dim3(1), dim3(64 * waves), 64 * 1024, 0, d, iters);
    hipLaunchKernelGGL((probe<MODE>), dim3(1), dim3(64 * waves), 64 * 1024, 0, d, iters);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // bytes read by the whole workgroup per cycle
    const double bytes = (double)waves * iters * 16 * 3 * 1024;
    printf("%-44s waves %d: %8.1f cycles per key tile per wave, %6.1f B/clk per CU\n", name, waves, (double)h[0] / (iters * 16.0), bytes / (double)h[0]);
}
int main() {
    unsigned long long *d;
    hipMalloc(&d, 8192 * 8);
    for (int waves : {1, 4, 8}) {
        if (waves == 1) { run<0>("2 x ds_read_b64 (current)", d, 1); run<1>("ds_read_b128 rows 176 B", d, 1); run<2>("ds_read_b128 linear", d, 1); }
        if (waves == 4) { run<0>("2 x ds_read_b64 (current)", d, 4); run<1>("ds_read_b128 rows 176 B", d, 4); run<2>("ds_read_b128 linear", d, 4); }
        if (waves == 8) { run<0>("2 x ds_read_b64 (current)", d, 8); run<1>("ds_read_b128 rows 176 B", d, 8); run<2>("ds_read_b128 linear", d, 8); }
    }
    return 0;
}
