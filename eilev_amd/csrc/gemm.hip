// gemm.hip — bf16 MFMA GEMM family for gfx950:  C[M,N] = epi(A[M,K] . W[N,K]^T + bias) (+ resid)
//
// Replaces every nn.Linear on the path (hf modeling_blip_2.py:328,351,366-368,584-586,618,660,674;
// hf modeling_opt.py:151-179,239-247,512; ref:eilev/model/v2.py:308).  Weights stay in the checkpoint's
// [out,in] layout: both MFMA operands then read 8 consecutive k per lane (16-byte loads), no transposes.
//
// Tiled kernel (M > 16): BMxBNx64 tile, 64-lane waves each owning a (BM/NWM)x(BN/NWN) sub-tile made of
// 16x16x32 MFMAs (v_mfma_f32_16x16x32_bf16, fp32 accumulate).  Global->register->LDS staging with the
// next tile's loads issued before the current tile's MFMAs (register prefetch) and a 2-deep LDS ring,
// one barrier per K-step.  LDS rows are 128 B (64 bf16); the 16-byte chunk index is XOR-swizzled with
// (row & 7) so the ds_read_b128 fragment reads of 16 consecutive rows spread over all bank groups.
// Workgroup ids are remapped so that each XCD (private L2) walks a contiguous range of tiles.
//
// Skinny kernel (M <= 16, the decode step): one MFMA row-block of 16 weight rows per workgroup, the K
// range split over the 8 waves (and over gridDim.y when N is small) — HBM-bound weight streaming.
//
// Files: gemm_common.h (tile map, swizzle, general epilogue), gemm_tiled.h (per-tile kernels), gemm_pp4.h (persistent ping-pong kernel),
// gemm_w6.h (one wave per SIMD), gemm_skinny.h (M <= 32 weight streaming); this file: the dispatch.  gemm_pp4_ext.hip holds the fp8 /
// LayerNorm-folding instances of the persistent kernel as a second object so that the two long hipcc runs go side by side (build.py).
#include "gemm_common.h"
#include "gemm_tiled.h"
#include "gemm_pp4.h"
#include "gemm_w6.h"
#include "gemm_skinny.h"

int g_eilev_grid_cus = 0;  // 0 = every CU (common.h: eilev_grid_cus); written by the probe build only
#ifdef EILEV_PROBES  // the probe build only (build.py --variant probes -DEILEV_PROBES): the product library has neither the switches nor their state
int g_gemm_debug = 0;  // probe-only switches (tools/gemm_probe.py): 1 = skip stores, 2 = skip main loop
extern "C" int eilev_debug_gemm_flags(int f) { g_gemm_debug = f; return 0; }
extern "C" int eilev_debug_grid_cus(int n) { g_eilev_grid_cus = n; return 0; }
#endif
int g_skinny_nb_default = 1;  // weight blocks per workgroup of the weight-streaming GEMV (set after measurement; see launch_gemm)
#ifdef EILEV_PROBES
unsigned long long *g_gemm_trace = nullptr;  // probe-only: see GemmArgs::trace
int g_gemm_trace_tiles = 0;
extern "C" int eilev_debug_gemm_trace(void *buf, int tiles) { g_gemm_trace = (unsigned long long *)buf; g_gemm_trace_tiles = tiles; return 0; }
#endif

namespace {

int launch_pp4(const GemmArgs &g, hipStream_t s) {
    static bool attr_set = false;
    static int num_cu = 0;
    constexpr int smem = PP4_SMEM;
    if (!attr_set) {
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        int dev = 0;
        EILEV_HIP_CHECK(hipGetDevice(&dev));
        EILEV_HIP_CHECK(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
        attr_set = true;
    }
    const int tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256);
    const int ncu = eilev_grid_cus() < num_cu ? eilev_grid_cus() : num_cu;
    const int grid = tiles < ncu ? tiles : ncu / 8 * 8;
    if (g.A8 || g.ln_rows || g.stat_out) return launch_pp4_ext(g, grid, s);  // fp8 MFMA / LayerNorm-folding instances (the other object)
    // 16 x 16 x 32 MFMAs where every tile of the launch is a whole lean tile (gemm_pp4.h: M16)
    if (pp4_all_lean(g) && g.epi != 1) {
        static bool attr16 = false;
        if (!attr16) {
            EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<0, false, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<2, false, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr16 = true;
        }
#ifndef EILEV_A3_MIN_K
#define EILEV_A3_MIN_K 5120  /* profiles/r06_a3_min_k.log: flan-t5-xl wo (N = 2048, K = 5120) +10 %, K = 4096 / 2560 / 2048 shapes -3 ... +1 % */
#endif
        if (g.epi == 0 && g.K >= EILEV_A3_MIN_K) {  // long K: the A operand three K-steps deep (gemm_pp4.h A3)
            static bool attr_a3 = false;
            if (!attr_a3) {
                EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<0, false, 0, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
                attr_a3 = true;
            }
            hipLaunchKernelGGL((gemm_pp4_kernel<0, false, 0, 2, true>), dim3(grid), dim3(512), smem, s, g);
        } else if (g.epi == 2) hipLaunchKernelGGL((gemm_pp4_kernel<2, false, 0, 2>), dim3(grid), dim3(512), smem, s, g);
        else hipLaunchKernelGGL((gemm_pp4_kernel<0, false, 0, 2>), dim3(grid), dim3(512), smem, s, g);
        EILEV_LAUNCH_CHECK();
        return EILEV_OK;
    }
    if (g.epi == 1) hipLaunchKernelGGL(gemm_pp4_kernel<1>, dim3(grid), dim3(512), smem, s, g);
    else if (g.epi == 2) hipLaunchKernelGGL(gemm_pp4_kernel<2>, dim3(grid), dim3(512), smem, s, g);
    else hipLaunchKernelGGL(gemm_pp4_kernel<0>, dim3(grid), dim3(512), smem, s, g);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

}  // namespace


static int launch_gemm_core(const GemmArgs &g_in, int prof_kind, hipStream_t s, bool *ln_done);

// GemmArgs::ln_out: the LayerNorm of the output rows is either produced by the split-K reduction of the decode GEMV (launch_gemm_core
// sets ln_done) or by a LayerNorm launch here.
// Would launch_gemm run this <= 32-row launch on gemm_rows32_kernel (the one kernel that reads / writes the row-block activation layout)?
bool gemm_rows32_takes(const GemmArgs &g) {
    if (!g.A || !g.W || !g.C || g.W8 || g.wscale || g.M <= 16 || g.M > 32 || g.K % 256 || g.patch_group || g.ln_rows || g.stat_out) return false;
    if ((g.lda & 7) || (g.ldw & 7) || ((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15) || (g.dbg & 268435456)) return false;
    SkinnyArgs a;
    a.g = g;
    a.mr = 32;
    int ks = 1, ksteps = 0, grid_x = 0;
    return rows32_plan(g, (g.N + 15) / 16, skinny_n_cu(), a, ks, ksteps, grid_x);
}

// include/eilev.h: eilev_stream_layout_pack
extern "C" int eilev_stream_layout_pack(const void *w, int64_t n, int64_t k, void *out, void *stream) {
    if (!w || !out || w == out || n < 1 || k < 1) return EILEV_E_BADARG;
    if (n > 0x7fffffff || k > 0x7fffffff || n * k > 0x7fffffff0ll) return EILEV_E_UNSUPPORTED;
    int ks = 0, ksteps = 0, grid_x = 0;
    if (!rows32_shape((int)n, (int)k, skinny_n_cu(), ks, ksteps, grid_x)) return EILEV_E_UNSUPPORTED;
    const int64_t chunks = n * (k >> 3);
    hipLaunchKernelGGL(stream_pack_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16 *)w, (bf16 *)out, (int)n, (int)k, grid_x);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

int launch_gemm(const GemmArgs &g, int prof_kind, hipStream_t s) {
    bool ln_done = false;
    const int rc = launch_gemm_core(g, prof_kind, s, &ln_done);
    if (rc != EILEV_OK || !g.ln_out || ln_done || g.M <= 0) return rc;
    if (g.out_f32 || g.patch_group) return EILEV_E_UNSUPPORTED;
    if (!g.ln_beta) return launch_rmsnorm(reinterpret_cast<const bf16 *>(g.C), g.ldc, g.ln_gamma, g.ln_out, g.N, g.M, g.N, g.ln_eps, s);  // T5: RMS
    return launch_layernorm(reinterpret_cast<const bf16 *>(g.C), g.ldc, g.ln_gamma, g.ln_beta, g.ln_out, g.N, g.M, g.N, g.ln_eps, s);
}

static int launch_gemm_core(const GemmArgs &g_in, int prof_kind, hipStream_t s, bool *ln_done) {
    GemmArgs g = g_in;
#ifdef EILEV_PROBES
    g.dbg = g_gemm_debug;
    g.trace = g_gemm_trace;
    g.trace_tiles = g_gemm_trace_tiles;
#else
    g.dbg = 0;  // (the product library: no process-global reaches a launch)
    g.trace = nullptr;
    g.trace_tiles = 0;
#endif
    if (g.dbg & 4096) g.lda = 0;   // probe: every A row aliases row 0 (cache-resident operand)
    if (g.dbg & 8192) g.ldw = 0;   // probe: every W row aliases row 0
    if (g.dbg & 131072) g.ldc = 0;  // probe: every output row aliases row 0 (stores stay in L2)
    if (g.M <= 0) return EILEV_OK;
    if (g.A8) {
        // fp8 activations x fp8 weights (eilev_linear_a8w8): the persistent ping-pong kernel on the fp8 MFMA, general epilogue
        // with the row and column scales
        if (!g.W8 || !g.C || !g.wscale || !g.ascale || g.N <= 0 || g.K <= 0) return EILEV_E_BADARG;
        if ((g.K % 128) || g.lda != g.K || g.ldw != g.K || ((uintptr_t)g.A8 & 15) || ((uintptr_t)g.W8 & 15) || g.patch_group != 0 ||
            (int64_t)g.M * g.K >= 0x7fff0000ll || (int64_t)g.N * g.K >= 0x7fff0000ll)
            return EILEV_E_UNSUPPORTED;
        if (!g.out_f32 && ((g.ldc & 7) || (g.N & 3) || ((uintptr_t)g.C & 15) || (g.resid && ((g.ldr & 7) || ((uintptr_t)g.resid & 15))))) return EILEV_E_UNSUPPORTED;
        if (prof_kind >= 0) prof_begin(prof_kind, 2.0 * g.M * (double)g.N * g.K, s);
        const int rc8 = launch_pp4(g, s);
        if (prof_kind >= 0) prof_end(s);
        return rc8;
    }
    if (g.W8) {
        // fp8 weights.  M <= 32 (decode): streamed as bytes by the fp8 skinny kernel.  Larger M (prefill): expanded to bf16 in the
        // caller's scratch (exact), then the bf16 kernels with the per-channel scale in their epilogue.
        if (!g.A || !g.C || !g.wscale || g.N <= 0 || g.K <= 0 || g.ldw != g.K || ((uintptr_t)g.W8 & 15) || (g.K & 15)) return EILEV_E_BADARG;
        if (!(g.M <= 32 && g.K % 256 == 0 && g.patch_group == 0)) {
            if (!g.w8_scratch) return EILEV_E_WORKSPACE;
            const int64_t n16 = (int64_t)g.N * g.K / 16;
            hipLaunchKernelGGL(w8_expand_kernel, dim3((unsigned)ceil_div64(n16, 256)), dim3(256), 0, s, g.W8, g.w8_scratch, n16);
            EILEV_LAUNCH_CHECK();
            GemmArgs e = g_in;
            e.W = g.w8_scratch;
            e.W8 = nullptr;
            return launch_gemm_core(e, prof_kind, s, ln_done);
        }
        g.W = reinterpret_cast<const bf16 *>(g.W8);  // (never dereferenced as bf16: the skinny fp8 kernel reads W8)
    }
    if (!g.A || !g.W || !g.C || g.N <= 0 || g.K <= 0) return EILEV_E_BADARG;
    if ((g.K & 7) || (g.lda & 7) || (g.ldw & 7) || ((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15)) return EILEV_E_UNSUPPORTED;
    const bool dma_ok = g.K % 256 == 0 && (!(g.dbg & 8) || g.W8);
    if (g.hm_tok && (!hm_takes(g) || (int64_t)g.M * g.lda * 2 >= 0x7fff0000ll)) return EILEV_E_UNSUPPORTED;  // head-major q|k|v: common.h hm_takes
    const bool ln_fold = g.ln_rows != nullptr || g.stat_out != nullptr;  // LayerNorm-folding variants: the persistent kernel only
    if (ln_fold && (g.W8 || g.wscale || g.out_f32 || g.patch_group || g.scale_cols || g.K % BK || (int64_t)g.N * g.ldw * 2 >= 0x7fff0000ll)) return EILEV_E_UNSUPPORTED;
    const bool skinny = (g.M <= 16 || (g.M <= 32 && dma_ok)) && g.patch_group == 0 && !ln_fold;
    if ((g.a_frag || g.c_frag || g.ln_frag) && !skinny) return EILEV_E_UNSUPPORTED;
    if (!skinny && !g.out_f32 && ((g.ldc & 7) || (g.N & 3) || ((uintptr_t)g.C & 15) || (g.resid && ((g.ldr & 7) || ((uintptr_t)g.resid & 15))) ||
                                   (g.bias && ((uintptr_t)g.bias & 7))))
        return EILEV_E_UNSUPPORTED;
    int rc;
    if (skinny) {
        SkinnyArgs a;
        a.g = g;
        a.mr = g.M <= 16 ? 16 : 32;
        const int nb = (g.N + 15) / 16;
        int ks = nb >= 384 ? 1 : (512 + nb - 1) / nb;  // >= 1.5 workgroups per CU: no split (and no reduce launch); (1024: decode 5.03 -> 5.18 ms/token)
        const int ksteps = (g.K + 31) / 32;
        if (ks > ksteps / 32) ks = ksteps / 32 > 0 ? ksteps / 32 : 1;  // >= 8 K-steps of 32 per wave
        if (ks > 1 && (!g.scratch || (size_t)ks * a.mr * g.N * sizeof(float) > g.scratch_bytes)) ks = 1;
        a.ks = ks;
        a.part = g.scratch;
        // tiles of 256 per wave: ceil(ceil(K / 256 / ks) / 4); up to 3 (every decode shape) the activations are preloaded
        const int per_w = ((g.K / 256 + ks - 1) / ks + 3) / 4;
        const bool pre = per_w <= 3 && !(g.dbg & 128);
        // weight blocks per workgroup (activation fragments reused): probe override (dbg >> 26) & 7 = 1 / 2 / 4; default by shape below
        int ks32 = 0, r32_grid = 0;
        int nbsel = (g.dbg >> 26) & 7;
        // measured at M = 32 (tools/skinny_sweep.py, 2 LDS stages so that two workgroups share a CU): lm_head (3142 blocks) 2.82 -> 3.45 /
        // 3.76 / 4.20 TB/s with 2 / 4 / 8 blocks per workgroup, qkv (480) 2.25 -> 2.46 with 2 (1.71 with 4: 120 workgroups leave CUs idle),
        // fc1 (640) 2.17 -> 2.30 with 2; the 2560-row matrices and M <= 16 are best with one block: keep >= 240 workgroups
        if (nbsel == 0) {
            nbsel = g_skinny_nb_default;
            // (workgroups = blocks x K splits: fc2 of OPT-2.7B has 160 blocks x 4 splits; probe flag 1 << 30: count blocks only, as before)
            const int wgs = (g.dbg & 1073741824) ? nb : nb * ks;
            if (g.M > 16) nbsel = wgs >= 8 * 240 ? 8 : (wgs >= 4 * 240 ? 4 : (wgs >= 2 * 240 ? 2 : 1));
        }
        if (nbsel == 7) nbsel = 8;  // probe encoding
        if (nbsel != 2 && nbsel != 4 && nbsel != 8) nbsel = 1;
        if (nbsel == 8 && g.M <= 16) nbsel = 4;
        if (g.W8) nbsel = 1;
        // gemm_rows32_kernel takes this launch?  (the row-block activation layouts exist in that kernel only)
        const bool r32 = !g.W8 && g.M > 16 && !(g.dbg & 268435456) && rows32_plan(g, nb, skinny_n_cu(), a, ks, ks32, r32_grid);
        if ((g.a_frag || g.c_frag || g.ln_frag) && !r32) return EILEV_E_UNSUPPORTED;
        if (g.W8 && g.M > 16) {
            if (pre) hipLaunchKernelGGL((gemm_skinny_w8_kernel<2, true>), dim3(nb, ks), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((gemm_skinny_w8_kernel<2, false>), dim3(nb, ks), dim3(256), 0, s, a);
        } else if (g.W8) {
            if (pre) hipLaunchKernelGGL((gemm_skinny_w8_kernel<1, true>), dim3(nb, ks), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((gemm_skinny_w8_kernel<1, false>), dim3(nb, ks), dim3(256), 0, s, a);
        } else if (r32) {
            // round 4 (gemm_rows32_kernel): one workgroup per CU, the 32 rows loaded once per CU.  probe flag 1 << 28: the kernels below
            // (the kernel deals the N weight rows over grid_x workgroups row by row; with a K split the grid is still one workgroup per CU)
            const int grid_x = r32_grid;
            if (a.g.Wp && (g.ldw != g.K || (g.dbg & (32768 | 134217728)))) a.g.Wp = nullptr;  // (the stream layout has no row stride; probe flags 1 << 15: ignore it, 1 << 27: first-fit plan — its grid may differ from the one the copy was dealt for)
            if (ks32 == 10) hipLaunchKernelGGL((gemm_rows32_kernel<2, 10, 3>), dim3(grid_x, ks), dim3(512), 0, s, a);
            else if (ks32 == 8) hipLaunchKernelGGL((gemm_rows32_kernel<2, 8, 3>), dim3(grid_x, ks), dim3(512), 0, s, a);
            else hipLaunchKernelGGL((gemm_rows32_kernel<2, 5, 4>), dim3(grid_x, ks), dim3(512), 0, s, a);
        } else if (dma_ok && pre && nbsel > 1) {
            // activations held across NB weight blocks per workgroup (see gemm_skinny_nb_kernel): 2 LDS-DMA stages + ping-pong partials =
            // 64 KB + 2 x MB x 4 KB, so two workgroups share a CU
            const int mbk = g.M > 16 ? 2 : 1;
            const int grid_x = (nb + nbsel - 1) / nbsel;
            const size_t sm = 4 * 2 * 8192 + (size_t)2 * 4 * mbk * 64 * 4 * 4;
            static bool attr_nb = false;
            if (!attr_nb) {
                const int mx = 4 * 2 * 8192 + 2 * 4 * 2 * 64 * 4 * 4;
                EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_skinny_nb_kernel<2, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, mx));
                EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_skinny_nb_kernel<2, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, mx));
                EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_skinny_nb_kernel<2, 8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, mx));
                EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_skinny_nb_kernel<1, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, mx));
                EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_skinny_nb_kernel<1, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, mx));
                attr_nb = true;
            }
            if (mbk == 2 && nbsel == 8) hipLaunchKernelGGL((gemm_skinny_nb_kernel<2, 8, 2>), dim3(grid_x, ks), dim3(256), sm, s, a);
            else if (mbk == 2 && nbsel == 4) hipLaunchKernelGGL((gemm_skinny_nb_kernel<2, 4, 2>), dim3(grid_x, ks), dim3(256), sm, s, a);
            else if (mbk == 2) hipLaunchKernelGGL((gemm_skinny_nb_kernel<2, 2, 2>), dim3(grid_x, ks), dim3(256), sm, s, a);
            else if (nbsel == 4) hipLaunchKernelGGL((gemm_skinny_nb_kernel<1, 4, 2>), dim3(grid_x, ks), dim3(256), sm, s, a);
            else hipLaunchKernelGGL((gemm_skinny_nb_kernel<1, 2, 2>), dim3(grid_x, ks), dim3(256), sm, s, a);
        } else if (dma_ok && g.M > 16) {
            if (pre) hipLaunchKernelGGL((gemm_skinny_dma_kernel<2, true>), dim3(nb, ks), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((gemm_skinny_dma_kernel<2, false>), dim3(nb, ks), dim3(256), 0, s, a);
        } else if (dma_ok) {
            if (pre) hipLaunchKernelGGL((gemm_skinny_dma_kernel<1, true>), dim3(nb, ks), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((gemm_skinny_dma_kernel<1, false>), dim3(nb, ks), dim3(256), 0, s, a);
        }
        else hipLaunchKernelGGL(gemm_skinny_kernel, dim3(nb, ks), dim3(256), 0, s, a);
        EILEV_LAUNCH_CHECK();
        if (ks > 1) {
            if (g.ln_out && !(g.dbg & 536870912) && !g.out_f32 && g.epi == 0 && g.scale_cols == 0 && (g.N & 7) == 0 && g.N <= 4096 && (g.ldc & 7) == 0 && (!g.resid || (g.ldr & 7) == 0)) {
                // split-K partials -> row (+ bias + residual) -> its LayerNorm in one launch (norm.hip)
                const int rc_ln = launch_reduce_ln(a.part, ks, a.mr, g.M, g.N, g.wscale, g.bias, g.resid, g.ldr, reinterpret_cast<bf16 *>(g.C), g.ldc,
                                                   g.ln_gamma, g.ln_beta, g.ln_out, g.ln_eps, s, g.ln_frag);
                if (rc_ln != EILEV_OK) return rc_ln;
                *ln_done = true;
                return EILEV_OK;
            }
            if (g.ln_frag) return EILEV_E_UNSUPPORTED;  // (the row-block LayerNorm rows exist in the fused reduce only)
            const int total = g.M * g.N;
            hipLaunchKernelGGL(skinny_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, a);
            EILEV_LAUNCH_CHECK();
        }
        return EILEV_OK;
    }
    // The LDS-DMA kernels address A through a 32-bit buffer offset: an A operand of 2 GiB or more (the Q-Former k|v
    // projection of a whole step: 1.1 M rows x 1408) is processed as row chunks that fit, each with the fast kernels
    const int64_t a_bytes = (int64_t)g.M * g.lda * 2;
    // (the persistent ping-pong kernel addresses A per tile: shapes it takes — >= 1024 tiles of 256 x 256 — are not chunked)
    const int64_t t256_pre = ceil_div64(g.M, 256) * ceil_div64(g.N, 256);
    const bool pp4_takes = g.K % BK == 0 && g.patch_group == 0 && t256_pre >= 1024 && (g.N >= 2048 || ceil_div64(g.M, 256) * ceil_div64(g.N, 128) >= 512) &&
                           (int64_t)g.N * g.ldw * 2 < 0x7fff0000ll && !g.dbg;
    if (a_bytes >= 0x7fff0000ll && g.K % BK == 0 && g.patch_group == 0 && !g.dbg && !pp4_takes) {
        const int64_t rows_per = (0x7fff0000ll / (g.lda * 2)) / 256 * 256;
        if (rows_per >= 256) {
            for (int64_t r0 = 0; r0 < g.M; r0 += rows_per) {
                GemmArgs c = g_in;
                c.M = (int)((g.M - r0) < rows_per ? (g.M - r0) : rows_per);
                c.A = g.A + r0 * g.lda;
                if (g.resid) c.resid = g.resid + r0 * g.ldr;
                if (g.stat_out) c.stat_out = g.stat_out + r0 * 2;  // stat_ld stays the row count of the whole matrix
                if (g.ln_rows) c.ln_rows = g.ln_rows + r0 * 2;
                c.C = g.out_f32 ? (void *)(reinterpret_cast<float *>(g.C) + r0 * g.ldc) : (void *)(reinterpret_cast<bf16 *>(g.C) + r0 * g.ldc);
                c.ln_out = nullptr;  // (the caller normalises the whole matrix once)
                const int rc_chunk = launch_gemm(c, prof_kind, s);
                if (rc_chunk != 0) return rc_chunk;
            }
            return EILEV_OK;
        }
    }
    const double flops = 2.0 * g.M * (double)g.N * g.K;
    if (ln_fold) {
        if (prof_kind >= 0) prof_begin(prof_kind, flops, s);
        const int rc_ln = launch_pp4(g, s);
        if (prof_kind >= 0) prof_end(s);
        return rc_ln;
    }
    if (prof_kind >= 0) prof_begin(prof_kind, flops, s);
    const int force = (g.dbg >> 4) & 15;  // probe-only override of the tile choice
    const int64_t tm256 = ceil_div64(g.M, 256);
    int cfg;
    if (tm256 * ceil_div64(g.N, 256) >= 256 && g.N >= 2048) cfg = 1;        // 256x256, 2 LDS stages, 1 WG/CU
    else if (tm256 * ceil_div64(g.N, 128) >= 512) cfg = 3;                   // 256x128, 1 stage, 2 WG/CU (N = 1408 / 1536)
    else if (tm256 * ceil_div64(g.N, 128) >= 192) cfg = 2;                   // 256x128, 2 stages
    else cfg = 4;                                                            // 128x128
    bool wide_tiles = false;
    const bool w6_ok = g.K % 64 == 0 && g.K >= 256 && g.N % 128 == 0 && (!g.resid || (g.epi == 0 && (g.ldr & 7) == 0)) && !g.out_f32 && g.patch_group == 0 && g.scale_cols == 0 && !g.wscale &&
                       (int64_t)g.M * g.lda * 2 < 0x7fff0000ll && (int64_t)g.N * g.ldw * 2 < 0x7fff0000ll && (g.ldc & 7) == 0;
    // N = 1408 / 1536 with >= 512 column-half tiles: the persistent kernels win at every row count measured (fc2 at 34 952 rows:
    // per-tile 696 us, ping-pong 641, one-wave-per-SIMD 582; at 279 616 rows the ping-pong kernel despite its N padding).  With
    // fewer tiles (192-511 halves: 17-34 frames) the one-wave-per-SIMD kernel alone wins (fc2 at 4369 rows: 104 -> 77 us; pp4 124)
    if (cfg == 3 && g.K % BK == 0 && !(g.dbg & 16384)) { cfg = 1; wide_tiles = true; }
    if (cfg == 2 && w6_ok && force == 0 && !(g.dbg & (16384 | 2097152 | 4))) { cfg = 1; wide_tiles = true; }
    if (force == 9) cfg = 1;  // probe: persistent kernel regardless of the shape
    if (force == 13 || force == 14 || force == 15) cfg = 4;  // probe: 64x128 / 128x128 tiles / split-K
    else if (force >= 1 && force <= 4) cfg = force;
    // one-wave-per-SIMD continuous-stream kernel (256 x 128 tiles): its smaller tiles balance better when there are fewer than
    // 4 rounds of 256 x 256 tiles (M = 7680 prefill GEMMs: +28 %); with more tiles the ping-pong kernel with the lean epilogue wins
    // (qkv +5 %, OPT out_proj +3 %)
    // Round 5 (profiles/r05_w6_vs_pp4_rows.log: rows swept 3840 .. 30 720 at N = 2048 / 2560 / 6144 / 7680 / 10 240): which of the two wins is
    // the wave quantisation of its tile count over the CUs — 256 x 256 tiles fill ceil(t / CUs) rounds, the 256 x 128 tiles of w6 twice as
    // many half-sized ones — times the ping-pong kernel's ~6 % higher rate at equal fill (e.g. flan-t5-xl wo / o at 30 720 rows: 960 tiles =
    // 3.75 rounds, ping-pong 1144 / 995 TFLOP/s against 1071 / 864; OPT qkv at 15 360 rows: 1800 tiles = 7.03 rounds, w6 1166 against 1101).
    // The old rule (w6 below 1024 tiles) stays for N that is not a whole number of 256-column tiles.
    const int64_t tiles256 = tm256 * ceil_div64(g.N, 256);
    const int n_cu_q = skinny_n_cu() / 8 * 8;
    auto fill = [&](int64_t t) { return (double)t / (double)(ceil_div64(t, n_cu_q) * n_cu_q); };
    const bool w6_pick = cfg == 1 && (g.N % 256 == 0 ? 1.06 * fill(tiles256) < fill(tm256 * ceil_div64(g.N, 128)) : tiles256 < 1024);
    if ((force == 12 || (force == 0 && w6_pick && !(g.dbg & (2097152 | 4)))) && w6_ok)
        rc = launch_w6(g, s);
    else if (cfg == 1 && !(wide_tiles && (g.dbg & 1048576)) && (force == 0 || force == 9) && !(g.dbg & 4) && g.K % BK == 0 && (int64_t)g.N * g.ldw * 2 < 0x7fff0000ll)
        rc = launch_pp4(g, s);  // persistent ping-pong kernel
    else if (cfg == 1) rc = launch_tiled<256, 256, 2, 4, 2, 2>(g, s);
    else if (cfg == 3) rc = launch_tiled<256, 128, 4, 2, 1, 4, 1>(g, s);
    else if (cfg == 2) rc = launch_tiled<256, 128, 4, 2, 2, 2>(g, s);
    else if (cfg == 4 && g.out_f32 && !g.bias && !g.resid && g.epi == 0 && !g.wscale && g.scale_cols == 0 && g.patch_group == 0 && g.K % BK == 0 &&
             g.K >= 8192 && ceil_div64(g.M, 128) * ceil_div64(g.N, 128) <= 128 && (force == 0 || force == 15) && !(g.dbg & 4) &&
             (int64_t)g.M * g.lda * 2 < 0x7fff0000ll && (int64_t)g.N * g.ldw * 2 < 0x7fff0000ll) {
        // weight-gradient shape (dW = dY^T X: a few output tiles, K = rows of the step): split K over enough slices to fill the CUs
        const int tiles = (int)(ceil_div64(g.M, 128) * ceil_div64(g.N, 128)), nk = g.K / BK;
        int slices = 512 / tiles;
        slices = slices < 2 ? 2 : (slices > 16 ? 16 : slices);
        if (slices > nk / 8) slices = nk / 8 > 1 ? nk / 8 : 1;
        GemmArgs gs = g;
        gs.k_slice = (nk + slices - 1) / slices;
        slices = (nk + gs.k_slice - 1) / gs.k_slice;
        static bool attr_set = false;
        constexpr int smem = 2 * (128 + 128) * 128;
        if (!attr_set) {
            EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_glds_kernel<128, 128, 2, 2, 0, 2, 2, 0>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr_set = true;
        }
        EILEV_HIP_CHECK(hipMemsetAsync(g.C, 0, (size_t)g.M * g.ldc * sizeof(float), s));
        hipLaunchKernelGGL((gemm_glds_kernel<128, 128, 2, 2, 0, 2, 2, 0>), dim3(tiles, slices), dim3(256), smem, s, gs);
        const hipError_t le = hipGetLastError();
        rc = le == hipSuccess ? EILEV_OK : (int)le;
    }
    else if ((force == 0 || force == 13) && cfg == 4 && (force == 13 || ceil_div64(g.M, 128) * ceil_div64(g.N, 128) < 96) && g.M > 64 && !(g.dbg & 4))
        rc = launch_tiled<64, 128, 1, 2, 2, 3>(g, s);  // a handful of 128x128 tiles (Q-Former graph: 544 rows): 64x128, 2 waves, 3 WG/CU (+13 % at 544 x 768 x 768; slower from ~160 tiles on)
    else rc = launch_tiled<128, 128, 2, 2, 2, 2>(g, s);
    if (prof_kind >= 0) prof_end(s);
    return rc;
}
