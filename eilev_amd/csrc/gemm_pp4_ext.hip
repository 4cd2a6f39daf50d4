// gemm_pp4_ext.hip — the fp8-MFMA and LayerNorm-folding instances of the persistent ping-pong kernel (gemm_pp4.h), called by launch_pp4 (gemm.hip)
// with its grid.  A separate object: these instances take as long to compile as the rest of the GEMM family together.
#include "gemm_pp4.h"

// The fp8-MFMA and LayerNorm-folding instances of the persistent kernel (called by launch_pp4 with its grid).
int launch_pp4_ext(const GemmArgs &g, int grid, hipStream_t s) {
    constexpr int smem = PP4_SMEM;
    if (g.A8 && g.hm_tok) return EILEV_E_UNSUPPORTED;
    if (g.A8) {  // fp8 x fp8 on the fp8 MFMA: byte operands, K halved so that the kernel's 2-byte strides are byte strides
        static bool attr8 = false;
        if (!attr8) {
            EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr8 = true;
        }
        if (g.epi == 1) return EILEV_E_UNSUPPORTED;
        GemmArgs h = g;
        h.A = reinterpret_cast<const bf16 *>(g.A8);
        h.W = reinterpret_cast<const bf16 *>(g.W8);
        h.K = g.K / 2; h.lda = g.lda / 2; h.ldw = g.ldw / 2;
        if (g.epi == 2) hipLaunchKernelGGL((gemm_pp4_kernel<2, true>), dim3(grid), dim3(512), smem, s, h);
        else hipLaunchKernelGGL((gemm_pp4_kernel<0, true>), dim3(grid), dim3(512), smem, s, h);
        EILEV_LAUNCH_CHECK();
        return EILEV_OK;
    }
    if (g.hm_tok && !hm_takes(g)) return EILEV_E_UNSUPPORTED;  // head-major q|k|v: the 16 x 16 folded-LayerNorm consumer only (common.h)
    // LayerNorm-folding variants: consumer (qkv, fc1 + GELU) / producer (proj, fc2 with the residual)
    static bool attr_ln = false;
    if (!attr_ln) {
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<0, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<1, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<0, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_ln = true;
    }
    if (g.epi == 2 || g.out_f32 || (g.ln_rows && (g.resid || g.stat_out || !g.ln_csum || ((uintptr_t)g.ln_csum & 15) || ((uintptr_t)g.ln_rows & 7))) ||
        (g.stat_out && (!g.resid || g.epi != 0 || g.stat_ld < g.M || ((uintptr_t)g.stat_out & 7))))
        return EILEV_E_UNSUPPORTED;
    if (g.stat_out && pp4_all_lean(g)) {  // statistics producers (ViT proj / fc2)
        static bool attr16s = false;
        if (!attr16s) {
            EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<0, false, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr16s = true;
        }
        hipLaunchKernelGGL((gemm_pp4_kernel<0, false, 2, 2>), dim3(grid), dim3(512), smem, s, g);
        EILEV_LAUNCH_CHECK();
        return EILEV_OK;
    }
    if (g.ln_rows && pp4_all_lean(g)) {  // folded-LayerNorm consumers (ViT qkv / fc1)
        static bool attr16 = false;
        if (!attr16) {
            EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<0, false, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<1, false, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr16 = true;
        }
        if (g.epi == 1) hipLaunchKernelGGL((gemm_pp4_kernel<1, false, 1, 1>), dim3(grid), dim3(512), smem, s, g);
        else hipLaunchKernelGGL((gemm_pp4_kernel<0, false, 1, 1>), dim3(grid), dim3(512), smem, s, g);
        EILEV_LAUNCH_CHECK();
        return EILEV_OK;
    }
    if (g.stat_out) hipLaunchKernelGGL((gemm_pp4_kernel<0, false, 2>), dim3(grid), dim3(512), smem, s, g);
    else if (g.epi == 1) hipLaunchKernelGGL((gemm_pp4_kernel<1, false, 1>), dim3(grid), dim3(512), smem, s, g);
    else hipLaunchKernelGGL((gemm_pp4_kernel<0, false, 1>), dim3(grid), dim3(512), smem, s, g);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}
