// cu_census.hip — which CUs does a (CU-masked) stream's work land on?  One workgroup per requested slot; lane 0 records XCC_ID and HW_ID and
// the workgroup then spins for `spin_us` so that concurrently launched workgroups spread over every CU the stream may use.
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/probes/libcu_census.so tools/probes/cu_census.hip   (tools/overlap_probe.py does it)
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void census_kernel(unsigned *out, long long spin_ticks) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID, bits [3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID, 32 bits
        out[blockIdx.x * 2] = xcc;
        out[blockIdx.x * 2 + 1] = hw;
    }
    const long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
}

// a streaming kernel that behaves like the decode weight stream: every workgroup reads its share of `bytes` with 16-byte loads
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_read_kernel(const u32x4_t *src, size_t n16, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const u32x4_t v = __builtin_nontemporal_load(src + i);
        acc ^= v[0] ^ v[3];
    }
    if (acc == 0x5ca1ab1eu) *sink = acc;
}

extern "C" int census_launch(unsigned *out, int blocks, int threads, int lds_bytes, int spin_us, void *stream) {
    hipLaunchKernelGGL(census_kernel, dim3(blocks), dim3(threads), lds_bytes, (hipStream_t)stream, out, (long long)spin_us * 100);
    return (int)hipGetLastError();
}
extern "C" int stream_read_launch(const void *src, size_t bytes, int blocks, unsigned *sink, void *stream) {
    hipLaunchKernelGGL(stream_read_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4_t *)src, bytes / 16, sink);
    return (int)hipGetLastError();
}
extern "C" int masked_stream_create(void **stream, const uint32_t *mask, int words) {
    hipStream_t s;
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
    *stream = (void *)s;
    return (int)e;
}
