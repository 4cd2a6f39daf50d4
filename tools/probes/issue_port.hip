// Probe: do MFMA and VALU / transcendental instructions of two waves on one SIMD overlap, or do they share one issue port?
// 512 threads = 8 waves per CU (2 per SIMD); waves 0-3 run stream A, waves 4-7 stream B (either may be empty).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
enum { NONE = 0, MFMA = 1, VALU = 2, TRANS = 3, MIX = 4, PKFMA = 5, PKMUL = 6 };
template <int KIND>
__device__ __forceinline__ float stream(int iters, float seed) {
    f32x4 acc[4] = {{seed, 0, 0, 0}, {0, seed, 0, 0}, {0, 0, seed, 0}, {0, 0, 0, seed}};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == MFMA || KIND == MIX) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
        }
        if constexpr (KIND == VALU || KIND == MIX) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[k]) : "v"(seed));
        }
        if constexpr (KIND == PKFMA || KIND == PKMUL) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 sv = {seed, seed};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    f2 x = {v[k], v[k + 1]};
                    if constexpr (KIND == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(sv));
                    else asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(sv));
                    v[k] = x.x; v[k + 1] = x.y;
                }
        }
        if constexpr (KIND == TRANS) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k]));
        }
    }
    float s = 0;
    for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    for (int k = 0; k < 8; ++k) s += v[k];
    return s;
}
template <int KA, int KB>
__global__ __launch_bounds__(512) void probe(float *out, int iters, float seed) {
    extern __shared__ char smem_force_one_wg_per_cu[];
    if (iters < 0) smem_force_one_wg_per_cu[threadIdx.x] = 1;
    const int wid = threadIdx.x >> 6;
    float r = 0;
    if (wid < 4) { if constexpr (KA != NONE) r = stream<KA>(iters, seed); }
    else { if constexpr (KB != NONE) r = stream<KB>(iters, seed); }
    if (r == 12345.678f) out[threadIdx.x] = r;
}
template <int KA, int KB>
float run(const char *name, float *d, int iters) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<KA, KB>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<KA, KB>), dim3(256), dim3(512), 100 * 1024, 0, d, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2 && ms < best) best = ms;
    }
    printf("%-28s %8.1f us\n", name, best * 1e3);
    return best;
}
int main() {
    float *d; hipMalloc(&d, 4096);
    const int it = 20000;  // per iteration: MFMA 4 x 16 = 64 cycles of matrix pipe; VALU 16 x 4 = 64 cycles; TRANS 8 x 16 = 128 cycles
    run<MFMA, NONE>("A: mfma        B: -", d, it);
    run<VALU, NONE>("A: valu        B: -", d, it);
    run<TRANS, NONE>("A: trans       B: -", d, it);
    run<MFMA, MFMA>("A: mfma        B: mfma", d, it);
    run<VALU, VALU>("A: valu        B: valu", d, it);
    run<MFMA, VALU>("A: mfma        B: valu", d, it);
    run<MFMA, TRANS>("A: mfma        B: trans", d, it);
    run<VALU, TRANS>("A: valu        B: trans", d, it);
    run<PKFMA, PKFMA>("A: pk_fma_f32  B: pk_fma_f32 (16 per iteration = 32 elements)", d, it);
    run<PKMUL, PKMUL>("A: pk_mul_f32  B: pk_mul_f32", d, it);
    run<PKFMA, NONE>("A: pk_fma_f32  B: -", d, it);
    run<MIX, NONE>("A: mfma+valu (one wave) B: -", d, it);
    run<MIX, MIX>("A: mfma+valu   B: mfma+valu", d, it);
    return 0;
}
