// grid_barrier.hip — the premise of a persistent decode block (round 5): what does a phase cost when the kernel boundary between two
// weight-streaming GEMVs is replaced by a device-wide barrier with the NEXT phase's weights requested before it?
//   mode 0: barrier only (256 workgroups x 512 threads, sense-reversing counter, agent-scope release / acquire)
//   mode 1: + every workgroup writes its 640 bytes of a 32 x 2560 bf16 row block and after the barrier reads the WHOLE block (164 KB,
//           the activations of the next GEMV) and checks it
//   mode 2: + every workgroup streams `wkb` KB of weights per phase into registers, requested BEFORE the barrier (prefetch across it)
//   mode 3: the same bytes as mode 2 as one launch per phase in a hipGraph (what the product does today)
// Build + run: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/grid_barrier tools/probes/grid_barrier.hip && gpurun_out/grid_barrier
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            printf("HIP error %d (%s) at line %d\n", (int)e_, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

struct Bar {
    unsigned count;
    unsigned pad0[31];
    unsigned gen;
    unsigned pad1[31];
    unsigned err;
};

// all threads call; returns false on a timeout (never hangs: the spin is bounded)
// FENCE = 1: agent-scope release / acquire fences (buffer_wbl2 sc1 / buffer_inv sc1: whole-L2 operations); FENCE = 0: none — the exchanged
// data itself is then written and read with device-scope (sc1) accesses
template <int FENCE>
__device__ __forceinline__ bool grid_barrier(Bar *b, unsigned G, unsigned &my_gen) {
    __builtin_amdgcn_s_waitcnt(0);  // this wave's stores have been acknowledged
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned old = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == G - 1) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int spins = 0;
            while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) {
                    ok = false;
                    atomicAdd(&b->err, 1u);
                    break;
                }
            }
        }
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    my_gen += 1;
    __syncthreads();
    return ok;
}

typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));
template <int NW, int FENCE>  // 16-byte weight loads per lane and phase (NW * 8 KB per workgroup)
__global__ __launch_bounds__(512) void phases_kernel(Bar *bar, int mode, int phases, uint16_t *act /* [2][32][2560] */, const u32x4_t *w, size_t w_phase16,
                                                      int w_bufs, unsigned *sink, unsigned *bad) {
    const unsigned G = gridDim.x;
    unsigned my_gen = 0;
    if (threadIdx.x == 0) my_gen = __hip_atomic_load(&bar->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    my_gen = __shfl(my_gen, 0, 64);
    __shared__ unsigned s_gen;
    if (threadIdx.x == 0) s_gen = my_gen;
    __syncthreads();
    my_gen = s_gen;
    unsigned acc = 0, nbad = 0;
    u32x4_t wv[NW > 0 ? NW : 1];
    auto request = [&](int p) {
        if (NW == 0) return;
        const u32x4_t *src = w + (size_t)(p % w_bufs) * w_phase16 + (size_t)blockIdx.x * (512 * NW) + threadIdx.x;
#pragma unroll
        for (int i = 0; i < NW; ++i) wv[i] = __builtin_nontemporal_load(src + i * 512);
    };
    if (mode >= 2) request(0);
    const __amdgpu_buffer_rsrc_t ract = __builtin_amdgcn_make_buffer_rsrc((void *)act, 0, 2 * 32 * 2560 * 2, 0x00020000);
    for (int p = 0; p < phases; ++p) {
        uint16_t *cur = act + (size_t)(p & 1) * (32 * 2560);
        if (mode >= 1) {  // this workgroup's 10 columns x 32 rows of the block: value = f(phase, row, col)
            if (threadIdx.x < 320) {
                const int r = threadIdx.x / 10, c = blockIdx.x * 10 + threadIdx.x % 10;
                const uint16_t val = (uint16_t)((p * 131 + r * 7 + c) & 0xffff);
                if (c < 2560) {
                    if (FENCE) cur[r * 2560 + c] = val;
                    else __builtin_amdgcn_raw_buffer_store_b16(val, ract, ((p & 1) * (32 * 2560) + r * 2560 + c) * 2, 0, 2 /* sc1 */);
                }
            }
        }
        if (!grid_barrier<FENCE>(bar, G, my_gen)) break;
        if (mode >= 1) {  // the whole block: 32 x 2560 x 2 bytes = 10240 16-byte chunks = 20 per thread
            const u32x4_t *a = reinterpret_cast<const u32x4_t *>(cur);
#pragma unroll 4
            for (int i = 0; i < 20; ++i) {
                const int idx = i * 512 + threadIdx.x;
                u32x4_t v;
                if (FENCE) v = a[idx];
                else v = __builtin_amdgcn_raw_buffer_load_b128(ract, ((p & 1) * (32 * 2560 / 8) + idx) * 16, 0, 2 /* sc1 */);
                const int e0 = idx * 8, r = e0 / 2560, c = e0 % 2560;
                const unsigned want = ((unsigned)((p * 131 + r * 7 + c) & 0xffff)) | ((unsigned)((p * 131 + r * 7 + c + 1) & 0xffff) << 16);
                nbad += v[0] != want;
                acc ^= v[1] ^ v[2] ^ v[3];
            }
        }
        if (mode >= 2) {
#pragma unroll
            for (int i = 0; i < NW; ++i) acc ^= wv[i][0] ^ wv[i][1] ^ wv[i][2] ^ wv[i][3];
            if (p + 1 < phases) request(p + 1);
        }
    }
    if (acc == 0x5ca1ab1eu) *sink = acc;
    if (nbad) atomicAdd(bad, nbad);
}

// mode 3: one launch per phase
// PAT 0: every wave instruction reads 1 KB contiguous.  PAT 1: the MFMA operand layout of gemm_rows32_kernel — lane (l15, lg) reads 16 bytes of
// row l15 at byte 640 wave + 64 u + 16 lg of a 5120-byte row: 16 segments of 64 bytes per instruction.  PAT 2: the same rows, but lane pairs
// cover a whole 128-byte line (lane (l15, lg) of instruction u reads bytes 128 (u / 2) + 32 lg + 16 (u & 1)): same lines, other order.
template <int NW, int PAT = 0>
__global__ __launch_bounds__(512) void one_phase_kernel(int p, uint16_t *act, const u32x4_t *w, size_t w_phase16, int w_bufs, unsigned *sink, unsigned *bad) {
    unsigned acc = 0, nbad = 0;
    u32x4_t wv[NW > 0 ? NW : 1];
    if (PAT == 0) {
        const u32x4_t *src = w + (size_t)(p % w_bufs) * w_phase16 + (size_t)blockIdx.x * (512 * NW) + threadIdx.x;
#pragma unroll
        for (int i = 0; i < NW; ++i) wv[i] = __builtin_nontemporal_load(src + i * 512);
    } else {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l15 = lane & 15, lg = lane >> 4;
        const char *base = reinterpret_cast<const char *>(w + (size_t)(p % w_bufs) * w_phase16) + (size_t)blockIdx.x * (NW / 10 * 16) * 5120;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int blk = i / 10, u = i % 10;
            const int byte = PAT == 1 ? wid * 640 + u * 64 + lg * 16 : wid * 640 + (u / 2) * 128 + lg * 32 + (u & 1) * 16;
            wv[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(base + (size_t)(blk * 16 + l15) * 5120 + byte));
        }
    }
    const uint16_t *prev = act + (size_t)((p + 1) & 1) * (32 * 2560);
    uint16_t *cur = act + (size_t)(p & 1) * (32 * 2560);
    const u32x4_t *a = reinterpret_cast<const u32x4_t *>(prev);
#pragma unroll 4
    for (int i = 0; i < 20; ++i) {
        const int idx = i * 512 + threadIdx.x;
        const u32x4_t v = a[idx];
        if (p > 0) {
            const int e0 = idx * 8, r = e0 / 2560, c = e0 % 2560;
            const unsigned want = ((unsigned)(((p - 1) * 131 + r * 7 + c) & 0xffff)) | ((unsigned)(((p - 1) * 131 + r * 7 + c + 1) & 0xffff) << 16);
            nbad += v[0] != want;
        }
        acc ^= v[1] ^ v[2] ^ v[3];
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) acc ^= wv[i][0] ^ wv[i][1] ^ wv[i][2] ^ wv[i][3];
    if (threadIdx.x < 320) {
        const int r = threadIdx.x / 10, c = blockIdx.x * 10 + threadIdx.x % 10;
        if (c < 2560) cur[r * 2560 + c] = (uint16_t)((p * 131 + r * 7 + c) & 0xffff);
    }
    if (acc == 0x5ca1ab1eu) *sink = acc;
    if (nbad) atomicAdd(bad, nbad);
}

template <int NW>
static void run(int G, int phases, int reps) {
    Bar *bar;
    uint16_t *act;
    u32x4_t *w;
    unsigned *sink, *bad;
    const size_t w_phase16 = (size_t)G * 512 * (NW > 0 ? NW : 1);
    const int w_bufs = 12;  // 12 distinct phase images: > the 256-MB Infinity Cache at 52 MB per phase
    CK(hipMalloc(&bar, sizeof(Bar)));
    CK(hipMemset(bar, 0, sizeof(Bar)));
    CK(hipMalloc(&act, 2 * 32 * 2560 * 2));
    CK(hipMemset(act, 0, 2 * 32 * 2560 * 2));
    CK(hipMalloc(&w, w_phase16 * 16 * w_bufs));
    CK(hipMemset(w, 1, w_phase16 * 16 * w_bufs));
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&bad, 4));
    CK(hipMemset(bad, 0, 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double mb = w_phase16 * 16 / 1e6;
    for (int fence = 1; fence >= 0; --fence)
    for (int mode = 0; mode <= 2; ++mode) {
        if (NW == 0 && mode == 2) continue;
        float best = 1e30f;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0, s));
            if (fence) hipLaunchKernelGGL((phases_kernel<NW, 1>), dim3(G), dim3(512), 0, s, bar, mode, phases, act, w, w_phase16, w_bufs, sink, bad);
            else hipLaunchKernelGGL((phases_kernel<NW, 0>), dim3(G), dim3(512), 0, s, bar, mode, phases, act, w, w_phase16, w_bufs, sink, bad);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        unsigned hb = 0, herr = 0;
        CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&herr, &bar->err, 4, hipMemcpyDeviceToHost));
        printf("NW=%2d (%.1f MB / phase) persistent %s mode %d: %.2f us / phase  (bad %u, barrier timeouts %u)", NW, mb, fence ? "fences" : "sc1   ", mode, best * 1e3 / phases, hb, herr);
        if (mode == 2) printf("  = %.2f TB/s", mb / (best * 1e3 / phases));
        printf("\n");
    }
    if constexpr (NW > 0) {  // one launch per phase in a graph
        hipGraph_t graph;
        hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int p = 0; p < phases; ++p)
            hipLaunchKernelGGL((one_phase_kernel<NW>), dim3(G), dim3(512), 0, s, p, act, w, w_phase16, w_bufs, sink, bad);
        CK(hipStreamEndCapture(s, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        float best = 1e30f;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(exec, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        unsigned hb = 0;
        CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("NW=%2d (%.1f MB / phase) one launch per phase (graph): %.2f us / phase = %.2f TB/s  (bad %u)\n", NW, mb, best * 1e3 / phases, mb / (best * 1e3 / phases), hb);
        if constexpr (NW % 10 == 0)
            for (int pat = 1; pat <= 2; ++pat) {
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                for (int p = 0; p < phases; ++p) {
                    if (pat == 1) hipLaunchKernelGGL((one_phase_kernel<NW, 1>), dim3(G), dim3(512), 0, s, p, act, w, w_phase16, w_bufs, sink, bad);
                    else hipLaunchKernelGGL((one_phase_kernel<NW, 2>), dim3(G), dim3(512), 0, s, p, act, w, w_phase16, w_bufs, sink, bad);
                }
                CK(hipStreamEndCapture(s, &graph));
                CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                best = 1e30f;
                for (int r = 0; r < reps; ++r) {
                    CK(hipEventRecord(e0, s));
                    CK(hipGraphLaunch(exec, s));
                    CK(hipEventRecord(e1, s));
                    CK(hipStreamSynchronize(s));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                printf("NW=%2d (%.1f MB / phase) one launch per phase, %s: %.2f us / phase = %.2f TB/s\n", NW, mb,
                       pat == 1 ? "MFMA-layout loads (16 x 64-byte segments per instruction)" : "line-pair order (16 x 2 x 32 bytes per instruction)", best * 1e3 / phases,
                       mb / (best * 1e3 / phases));
            }
    }
    fflush(stdout);
    CK(hipFree(w));
}

int main(int argc, char **argv) {
    int G = 256;
    int n_cu = 0;
    CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    if (n_cu > 0) G = n_cu;
    const int phases = argc > 1 ? atoi(argv[1]) : 128, reps = 5;
    printf("grid %d workgroups x 512 threads, %d phases, best of %d\n", G, phases, reps);
    run<0>(G, phases, reps);
    run<6>(G, phases, reps);    // 13 MB per phase (out_proj)
    run<19>(G, phases, reps);   // 40 MB (q|k|v)
    run<25>(G, phases, reps);   // 52 MB (fc1 / fc2)
    run<20>(G, phases, reps);   // 42 MB: 2 blocks of 16 rows per workgroup
    run<30>(G, phases, reps);   // 63 MB: 3 blocks
    return 0;
}
