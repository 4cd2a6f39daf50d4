// Included by attention.hip (inside its anonymous namespace, after attn_frame_kernel).
//
// Frame attention, two wave groups one phase apart: S = 257 exactly (ViT-g: 256 patches + CLS), hd = 88.  Same LDS images, fragments and
// exact single-pass softmax as attn_frame_kernel; what changes is who computes what, and when.
//  * 8 waves (two per SIMD, 256 VGPRs each); wave w owns query tiles w and w + 8 and runs them TOGETHER: one K (V) fragment read feeds
//    the MFMAs of both tiles (half the LDS traffic of one tile at a time: attn_frame_kernel spends 12.5 k of its 25 k cycles per pair
//    on LDS reads), S^T of both (2 x 68 fp32) stays in registers;
//  * query row 256 is split over the KEYS: wave w scores it against key tiles 2w, 2w + 1 (wave 0 also tile 16), keeps a partial softmax
//    with its own maximum, leaves (max, sum, partial O[96]) in a 4-KiB LDS scratch; every wave finishes 12 of the row's 88 outputs from
//    the eight partials in a fixed order;
//  * the round-2 form of this split ran its 8 waves in lock-step (S, softmax, PV phases paid one after the other: 584 us against
//    500 us per 544 frames).  Here waves 0-3 (group E, one per SIMD) and waves 4-7 (group L) execute DIFFERENT code between the same
//    three workgroup barriers of a pair — L runs one slot behind E:
//        slot 0   E: S(p)                 L: PV(p-1), stores, CLS partial
//        slot 1   E: softmax(p)           L: S(p)
//        slot 2   E: PV(p), stores, ...   L: softmax(p)
//    so a SIMD's MFMA / LDS phases of one wave run under the VALU phase of the other;
//  * K(p) is read in slots 0-1 of pair p and V(p) in slot 2 of p and slot 0 of p + 1: never more than two images are live, so three
//    LDS buffers serve as a ring over the image sequence K(0) V(0) K(1) V(1) ... (image n in buffer n % 3): V(p+1) is DMA'd from the
//    third barrier of pair p (three slots ahead of its first use), K(p+1) from the second barrier of pair p (two slots ahead);
//  * vector-memory operations of a wave retire in order, so "my pieces of image X have landed" is a counted s_waitcnt vmcnt(N) with
//    N = the operations issued after them (the counts below are lower bounds: the conditional CLS store is not counted).
template <class F, int... I>
__device__ __forceinline__ void attn_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void attn_static_for(F &&f) {
    attn_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// HM (round 5): q, k and v of a (frame, head) are each one block of [S][64] followed by [S][HD - 64] elements (what the q|k|v GEMM
// writes with GemmArgs::hm_tok): an image is staged from two contiguous runs, the LDS image ([key][HD], compact) is unchanged.
template <int HD, int NT, bool HM = false>
__global__ __launch_bounds__(512) void attn_frame3_kernel(const AttnArgs a) {
    constexpr int NW = 8;
    constexpr int CH = HD / 8, RS = HD * 2;
    constexpr int NPIECE = (NT * 16 * CH + (12 - CH) + 63) / 64;
    constexpr int BUF = NPIECE * 1024;
    constexpr int PPW = (NPIECE + NW - 1) / NW;
    constexpr int KS = 3, NKS = (NT + 1) / 2, DT = 6;
    constexpr int SCR = 3 * BUF;  // [NW][128] floats: O[0..95], max * scale * log2 e, sum
    static_assert(NT == 2 * NW + 1 && HD <= 96 && HD % 8 == 0 && NT * 16 * RS < 65536 - 512, "tile split / ds offset field");
    static_assert(PPW == 6, "the counted s_waitcnt below assume 6 DMA pieces per wave and image");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char_t *lds = (lds_char_t *)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int S = a.sq;  // == 16 * (NT - 1) + 1 (launcher)
    const int npairs = a.batch * a.heads;
    const float sl2 = a.scale * 1.44269504088896340736f;

    // DMA piece k (of PPW) of this wave: 64 lanes x 16 bytes of the image at src -> LDS buffer buf.  Issued one at a time BETWEEN compute
    // steps: 48 pieces (+ 48 Q loads) issued together at a barrier stall every wave at the texture-address unit for 3-4 k cycles.
    // (r4) the chunk -> (key, 16-byte column) split: piece i starts at chunk 64 i = CH q + r with q, r on the SCALAR unit; lane l's chunk is
    // key q + (r + l) / CH with (r + l) < 64 + CH divided by a 24-bit multiply and a shift (DIVM, checked below), and the row offset is a
    // 24-bit multiply-add: 5 full-rate VALU instructions per piece where pch / CH, key * ld2 compiled to v_mul_hi_i32 + v_mad_u64_u32 +
    // v_mul_lo_u32 (quarter rate: ~72 issue cycles per piece, 24 pieces per SIMD and pair = 12 % of the issue budget of a pair).
    // Rows past S - 1 are outside the descriptor's range and read zeros (masked keys of the last tile; V rows that meet P = 0).
    constexpr int DIVS = 9, DIVM = ((1 << DIVS) + CH - 1) / CH;
    static_assert([] {
        for (int t = 0; t < 64 + CH; ++t)
            if (((t * DIVM) >> DIVS) != t / CH) return false;
        return true;
    }(), "chunk / CH by multiply-shift");
    auto stage_piece = [&](const bf16 *src, int buf, int k) {
        const int ld2 = __builtin_amdgcn_readfirstlane((int)(a.ldk * 2));
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, HM ? S * HD * 2 : (S - 1) * ld2 + HD * 2, 0x00020000);
        int ln = lane;
        asm volatile("" : "+v"(ln));  // opaque: the piece geometry is recomputed per call (a handful of VALU ops), not kept in registers
        int i = wid + NW * k;
        i = i < NPIECE ? i : NPIECE - 1;
        const int q0 = (i * 64) / CH, r0 = i * 64 - q0 * CH;  // wave-uniform
        const unsigned t = (unsigned)(ln + r0);
        const unsigned kq = __umul24(t, DIVM) >> DIVS;
        if constexpr (HM) {
            // chunk c = t - CH kq of key q0 + kq: c < 8 in the [S][64] block (128-byte rows), else in the [S][HD - 64] block behind it;
            // keys past S - 1 (zeros in the LDS image) would land in the second block: sent outside the descriptor instead
            const unsigned key = (unsigned)q0 + kq, c = t - __umul24(kq, CH);
            const unsigned va = (key << 7) + (c << 4), vb = (unsigned)(S * 128 - 128) + __umul24(key, (HD - 64) * 2) + (c << 4);  // vb: (c - 8) * 16
            unsigned voff = c < 8 ? va : vb;
            voff = key < (unsigned)S ? voff : 0x7ffffff0u;
            attn_dma16(r, (lds_void_t *)(smem + buf * BUF + i * 1024), voff);
            return;
        }
        // (q0 + kq) ld2 + 16 (t - CH kq) = kq (ld2 - 16 CH) + (q0 ld2 + 16 t): no remainder to form
        const unsigned voff = __umul24(kq, (unsigned)(ld2 - 16 * CH)) + ((unsigned)(q0 * ld2) + (t << 4));  // kq < 2^9, ld2 < 2^24
        attn_dma16(r, (lds_void_t *)(smem + buf * BUF + i * 1024), voff);
    };
    auto stage = [&](const bf16 *src, int buf) {
#pragma unroll
        for (int k = 0; k < PPW; ++k) stage_piece(src, buf, k);
    };
    auto k_base = [&](int pair) -> const bf16 * {
        const int b = pair / a.heads, h = pair - b * a.heads;
        return a.k + (int64_t)b * a.k_bs + (int64_t)h * a.k_hs;
    };
    auto v_base = [&](int pair) -> const bf16 * {
        const int b = pair / a.heads, h = pair - b * a.heads;
        return a.v + (int64_t)b * a.v_bs + (int64_t)h * a.v_hs;
    };
    auto stage_k = [&](int pair, int buf) { stage(k_base(pair), buf); };
    auto stage_v = [&](int pair, int buf) { stage(v_base(pair), buf); };
    // Q fragments (3 x 16 bytes per lane) of tiles w, w + 8 and 16.  The loads are inline asm and so are their waits: hipcc does not count
    // the LDS-DMA pieces in flight when it places the s_waitcnt of a register load (its vmcnt(N) comes out N too small by the number of
    // pieces issued after the load, i.e. it drains the pieces), which would pull every image's deadline forward to the next Q use.
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_q;
    u32x2_q qlo[3][KS], qhi[3][KS];
    auto load_q = [&](int pair, int tile, u32x2_q (&lo)[KS], u32x2_q (&hi)[KS]) {
        const int b = pair / a.heads, h = pair - b * a.heads;
        int l15o = l15, go = g;
        asm volatile("" : "+v"(l15o), "+v"(go));  // (opaque: see stage)
        const int row = tile * 16 + l15o;
        // one descriptor per (frame, head): rows past S and head-dim slots past HD read 0 through the bounds / the offset trick below
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)(a.q + (int64_t)b * a.q_bs + (int64_t)h * a.q_hs), 0,
                                                                            HM ? S * HD * 2 : (int)(((int64_t)(S - 1) * a.ldq + HD) * 2), 0x00020000);
        const unsigned rb = row < S ? (HM ? (unsigned)row << 7 : __umul24((unsigned)row, (unsigned)(a.ldq * 2))) : 0x7ffffff0u;  // out of bounds -> 0 (row < 2^9, ldq < 2^23)
        // HM: head dims 64 .. HD - 1 of a row live in the second block ([S][HD - 64] behind the [S][64] block)
        const unsigned rb2 = row < S ? (unsigned)(S * 128) + __umul24((unsigned)row, (HD - 64) * 2) - 128u : 0x7ffffff0u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = ks * 32 + 4 * go, d1 = d0 + 16;  // go < 4: only a slot whose LAST lane group passes HD needs the per-lane test
            const unsigned base = (HM && ks >= 2) ? rb2 : rb;  // (rb2 + d * 2 with d >= 64: the - 128 above)
            const unsigned o0 = (ks * 32 + 16 <= HD || d0 + 4 <= HD) ? base + d0 * 2 : 0x7ffffff0u;
            const unsigned o1 = (ks * 32 + 32 <= HD || d1 + 4 <= HD) ? base + d1 * 2 : 0x7ffffff0u;
            asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(lo[ks]) : "v"(o0), "s"(rq));
            asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(hi[ks]) : "v"(o1), "s"(rq));
        }
    };
    auto load_q01 = [&](int pair) {  // 12 vector loads
        load_q(pair, wid, qlo[0], qhi[0]);
        load_q(pair, wid + NW, qlo[1], qhi[1]);
    };
    auto load_qc = [&](int pair) { load_q(pair, NT - 1, qlo[2], qhi[2]); };  // 6 vector loads
    // "at most N younger vector-memory operations outstanding"; the fragment registers are in / out operands so that nothing reads them earlier
#define FA_QWAIT(n, t)                                                                                                                 \
    asm volatile("s_waitcnt vmcnt(" #n ")"                                                                                             \
                 : "+v"(qlo[t][0]), "+v"(qlo[t][1]), "+v"(qlo[t][2]), "+v"(qhi[t][0]), "+v"(qhi[t][1]), "+v"(qhi[t][2])::"memory")
    // the same with the count chosen by a wave-uniform flag INSIDE one statement (an if / else of two statements makes hipcc copy the
    // still-pending registers into the registers the two branches agree on, before the wait)
#define FA_QWAIT2(n1, n0, flag, t)                                                                                                     \
    asm volatile("s_cmp_eq_u32 %6, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(" #n1 ")\n\ts_branch 2f\n1:\n\ts_waitcnt vmcnt(" #n0 ")\n2:" \
                 : "+v"(qlo[t][0]), "+v"(qlo[t][1]), "+v"(qlo[t][2]), "+v"(qhi[t][0]), "+v"(qhi[t][1]), "+v"(qhi[t][2])               \
                 : "s"(__builtin_amdgcn_readfirstlane((int)(flag)))                                                                                                  \
                 : "memory", "scc")
    auto q_frag = [&](int t, bf16x8 (&dst)[KS]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x4 l4 = __builtin_bit_cast(bf16x4, qlo[t][ks]), h4 = __builtin_bit_cast(bf16x4, qhi[t][ks]);
            dst[ks] = (bf16x8){l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
        }
    };
#define FA_BARRIER()                       \
    do {                                   \
        __builtin_amdgcn_sched_barrier(0); \
        __builtin_amdgcn_s_barrier();      \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)
#define FA_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
    auto s_issue = [&](unsigned kaddr, bf16x4 (&kf)[2 * KS], auto t_c) {
        constexpr int t = decltype(t_c)::value;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(kf[ks * 2 + hf]) : "v"(kaddr), "i"(t * 16 * RS + ks * 64 + hf * 32));
    };
    // wait for the fragments of one key tile (NAFTER younger ds_read_b64 may stay in flight), then one MFMA chain per query tile
    auto k_wait = [&](bf16x4 (&kf)[2 * KS], auto nafter) {
        constexpr int NAFTER = decltype(nafter)::value;
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]), "+v"(kf[5]) : "n"(NAFTER));
    };
    auto s_mma = [&](const bf16x8 (&q)[KS], const bf16x4 (&kf)[2 * KS]) -> f32x4 {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x4 lo = kf[ks * 2], hi = kf[ks * 2 + 1];
            const bf16x8 k0 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, q[ks], acc, 0, 0, 0);
        }
        return acc;
    };
    auto mask_last = [&](f32x4 &v) {  // keys of tile NT - 1 that do not exist (A row 4 g + r -> key)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ar = 4 * g + r;
            if ((NT - 1) * 16 + (ar < 8 ? 2 * ar : 2 * ar - 15) >= S) v[r] = -1e30f;
        }
    };
    // Reductions over lanes l, l ^ 16, l ^ 32, l ^ 48 through v_permlane16_swap / v_permlane32_swap of two copies (after the swap one copy
    // holds the own row / half, the other the partner's): six VALU instructions where two __shfl_xor are two ds_bpermute round trips behind
    // s_waitcnt lgkmcnt(0) — twelve of them per wave and pair, inside the VALU-bound softmax phase (r4).  Same operands, same sums.
    // Inline asm: hipcc -O3 folds the second result of __builtin_amdgcn_permlane16_swap into the first when both feed one max / add
    // (ROCm 7.2: `fadd %p, %p` in the IR), and fmaxf on asm results adds two canonicalising v_max.  The s_nop mirror the wait states hipcc
    // itself places between a VALU write and a lane swap that reads it.
#define FA_QUAD(OP, x)                                                                                                                   \
    do {                                                                                                                                 \
        float fa_b_ = (x);                                                                                                               \
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\t" OP " %0, %0, %1\n\tv_mov_b32 %1, %0\n\ts_nop 1\n\t"                  \
                     "v_permlane32_swap_b32 %0, %1\n\t" OP " %0, %0, %1"                                                                 \
                     : "+v"(x), "+v"(fa_b_));                                                                                            \
    } while (0)
    auto quad_max = [&](float x) -> float {
        FA_QUAD("v_max_f32", x);
        return x;
    };
    auto quad_sum = [&](float x) -> float {
        FA_QUAD("v_add_f32", x);
        return x;
    };
    auto row_offset = [&](f32x4 (&sc)[NT]) -> float {
        mask_last(sc[NT - 1]);
        float m4[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
#pragma unroll
        for (int t = 0; t < NT; t += 2)
#pragma unroll
            for (int r = 0; r < 4; ++r) m4[r] = t + 1 < NT ? fmaxf(fmaxf(m4[r], sc[t][r]), sc[t + 1][r]) : fmaxf(m4[r], sc[t][r]);
        return -quad_max(fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]))) * sl2;
    };
    auto sm_tile = [&](const f32x4 &sv, bf16x4 &pv, float nm, f32x2 &la, f32x2 &lb) {
        const f32x2 x0 = (f32x2){sv[0], sv[1]} * sl2 + nm, x1 = (f32x2){sv[2], sv[3]} * sl2 + nm;
        const f32x2 p0 = {__builtin_amdgcn_exp2f(x0.x), __builtin_amdgcn_exp2f(x0.y)};
        const f32x2 p1 = {__builtin_amdgcn_exp2f(x1.x), __builtin_amdgcn_exp2f(x1.y)};
        la += p0;
        lb += p1;
        pv = (bf16x4){(bf16)p0.x, (bf16)p0.y, (bf16)p1.x, (bf16)p1.y};
    };
    auto row_sum = [&](f32x2 la, f32x2 lb) -> float {
        return quad_sum((la.x + la.y) + (lb.x + lb.y));
    };
    auto softmax = [&](f32x4 (&sc)[NT], bf16x4 (&pr)[NT]) -> float {
        const float nm = row_offset(sc);
        f32x2 la = {0.f, 0.f}, lb = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            sm_tile(sc[t], pr[t], nm, la, lb);
            asm volatile("" : "+v"(pr[t]));  // P is rounded HERE: left alone, hipcc carries the fp32 values across the barrier (2x the registers) and converts in the PV phase
        }
        return row_sum(la, lb);
    };
    auto pv_issue = [&](unsigned va, bf16x4 (&vf)[2 * DT], auto k2_c, auto two_c) {
        constexpr int k2 = decltype(k2_c)::value;
        constexpr bool two = decltype(two_c)::value;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vf[2 * dt]) : "v"(va), "i"(2 * k2 * 16 * RS + dt * 32));
            if constexpr (two)
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vf[2 * dt + 1]) : "v"(va), "i"((2 * k2 + 1) * 16 * RS + dt * 32));
        }
    };
    auto v_wait = [&](bf16x4 (&vf)[2 * DT]) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]), "+v"(vf[4]), "+v"(vf[5]), "+v"(vf[6]), "+v"(vf[7]), "+v"(vf[8]),
                       "+v"(vf[9]), "+v"(vf[10]), "+v"(vf[11]));
    };
    auto pv_mma = [&](f32x4 (&o)[DT], bf16x4 p0, bf16x4 p1, const bf16x4 (&vf)[2 * DT], auto two_c) {
        constexpr bool two = decltype(two_c)::value;
        constexpr bf16x4 z4 = {0, 0, 0, 0};
        const bf16x8 pb = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const bf16x4 lo = vf[2 * dt], hi = two ? vf[2 * dt + 1] : z4;
            const bf16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v8, pb, o[dt], 0, 0, 0);
        }
    };
    auto store_o = [&](const f32x4 (&o)[DT], float l, bf16 *ob, int tile) {  // 3 vector stores
        int l15o = l15, go = lane >> 4;
        asm volatile("" : "+v"(l15o), "+v"(go));  // (opaque: see stage)
        const int row = tile * 16 + l15o;
        const float inv = __builtin_amdgcn_rcpf(l);  // l >= 1 (the row's maximum contributes exp2(0)); 1 ulp, the output is bf16
        const f32x2 inv2 = {inv, inv};
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)ob, 0, (int)(((int64_t)(S - 1) * a.ldo + HD) * 2), 0x00020000);
        const unsigned ob_off = __umul24((unsigned)row, (unsigned)(a.ldo * 2)) + ((go & 1) * 16 + (go >> 1) * 8) * 2;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_o;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            union { bf16x4 v; unsigned w[2]; } e, odd, lo, hi;
#pragma unroll
            for (int w = 0; w < 2; ++w) {  // (the accumulators' even-aligned register pairs: v_pk_mul_f32 + v_cvt_pk_bf16_f32 per dword)
                const f32x2 ev = (f32x2){o[2 * m][2 * w], o[2 * m][2 * w + 1]} * inv2, ov = (f32x2){o[2 * m + 1][2 * w], o[2 * m + 1][2 * w + 1]} * inv2;
                e.v[2 * w] = (bf16)ev.x;
                e.v[2 * w + 1] = (bf16)ev.y;
                odd.v[2 * w] = (bf16)ov.x;
                odd.v[2 * w + 1] = (bf16)ov.y;
            }
            // lanes g and g ^ 1 (16 lanes apart) trade halves: even g keeps its even tile and takes the partner's, odd g the odd tiles.
            // v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of its second: one VALU
            // instruction per dword where a ds_bpermute round trip (~100 cycles each, 12 per wave and pair) and two selects were
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const auto sw = __builtin_amdgcn_permlane16_swap(e.w[w], odd.w[w], false, false);
                lo.w[w] = sw[0];
                hi.w[w] = sw[1];
            }
            const int d0 = 32 * m + (go & 1) * 16 + (go >> 1) * 8;
            const bf16x8 v8 = {lo.v[0], lo.v[1], lo.v[2], lo.v[3], hi.v[0], hi.v[1], hi.v[2], hi.v[3]};
            // rows past S and d past HD are dropped by the offset (out of the descriptor's range)
#ifdef EILEV_PROBES
            if (a.dbg & 2048) continue;  // TIMING PROBE (no output): the kernel without its O stores
#endif
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_o, v8), ro, (row < S && d0 + 8 <= HD) ? ob_off + 64 * m : 0x7ffffff0u, 0, 0);
        }
    };
    // the eight partials of query row S - 1 -> its output row (every wave 12 of the outputs; the partials were written before the barrier just passed)
    auto cls_merge = [&](bf16 *obp) {
        int ln = threadIdx.x & 63;
        asm volatile("" : "+v"(ln));  // (opaque: see stage)
        const float *sp = reinterpret_cast<const float *>(smem + SCR);
        const int d = wid * 12 + (ln < 12 ? ln : 0);  // 8 waves x 12 = 96 >= HD
        float M = -3.0e38f;
#pragma unroll
        for (int w = 0; w < NW; ++w) M = fmaxf(M, sp[w * 128 + 96]);
        float L = 0.f, o0 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float f = __builtin_amdgcn_exp2f(sp[w * 128 + 96] - M);
            L = fmaf(sp[w * 128 + 97], f, L);
            o0 = fmaf(sp[w * 128 + d], f, o0);
        }
        if (ln < 12 && d < HD) obp[(int64_t)(S - 1) * a.ldo + d] = (bf16)(o0 / L);
    };
    auto out_base = [&](int pair) -> bf16 * {
        const int b = pair / a.heads, h = pair - b * a.heads;
        return a.o + (int64_t)b * a.o_bs + (int64_t)h * a.o_hs;
    };
    using I0 = std::integral_constant<int, 0>;

    // ---- the three phases of a pair, on this wave's registers -------------------------------------------------------------
    f32x4 scA[NT], scB[NT];
    bf16x4 prA[NT], prB[NT], prC[3];
    float lA, lB, mC, lC;
    // S^T of both tiles and of this wave's share of the CLS row; K(pair) is in LDS buffer kb
    auto phase_s = [&](int kb, bool q18, bool qc6, auto &&tile_hook) {
        unsigned kaddr;
        {   // (recomputed per pair from the lane id: kept across the loop it is the first thing hipcc spills)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int l15o = ln & 15, go = ln >> 4;
            kaddr = (unsigned)(uintptr_t)lds + kb * BUF + (l15o < 8 ? 2 * l15o : 2 * l15o - 15) * RS + go * 8;
        }
        {   // S^T of both tiles: every key tile's fragments are read once
            bf16x4 kf[2][2 * KS];
            s_issue(kaddr, kf[0], I0{});
            // Q of tiles w, w + 8: younger than them 6 pieces of V, 6 stores, 6 loads of the CLS row's Q (E) / 6 stores, 6 loads (L), and
            // in group L the 6 pieces of K(p+1) when they were issued
            FA_QWAIT2(18, 12, q18, 0);
            FA_QWAIT2(18, 12, q18, 1);
            bf16x8 qa[KS], qb[KS];
            q_frag(0, qa);
            q_frag(1, qb);
            attn_static_for<NT>([&](auto t_c) {
                constexpr int t = decltype(t_c)::value;
                if constexpr (t + 1 < NT) {
                    s_issue(kaddr, kf[(t + 1) & 1], std::integral_constant<int, t + 1>{});
                    k_wait(kf[t & 1], std::integral_constant<int, 2 * KS>{});
                } else {
                    k_wait(kf[t & 1], I0{});
                }
                scA[t] = s_mma(qa, kf[t & 1]);
                scB[t] = s_mma(qb, kf[t & 1]);
                tile_hook(t_c);
            });
        }
        {   // query row S - 1 against this wave's keys: tiles 2w, 2w + 1 (wave 0: + tile NT - 1)
            f32x4 scC[3];
            bf16x4 kf[2 * KS];
            const unsigned kaddr_c = kaddr + (unsigned)wid * (2 * 16 * RS);
            FA_QWAIT2(6, 0, qc6, 2);  // group L: the pieces of K(p+1) are younger
            bf16x8 qc[KS];
            q_frag(2, qc);
            s_issue(kaddr_c, kf, I0{});
            k_wait(kf, I0{});
            scC[0] = s_mma(qc, kf);
            s_issue(kaddr_c, kf, std::integral_constant<int, 1>{});
            k_wait(kf, I0{});
            scC[1] = s_mma(qc, kf);
            scC[2] = (f32x4){-1e30f, -1e30f, -1e30f, -1e30f};
            if (wid == 0) {
                s_issue(kaddr, kf, std::integral_constant<int, NT - 1>{});
                k_wait(kf, I0{});
                scC[2] = s_mma(qc, kf);
                mask_last(scC[2]);
            }
            float mx = -1e30f;
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, scC[u][r]);
            mC = quad_max(mx) * sl2;
            f32x2 la = {0.f, 0.f}, lb = {0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                sm_tile(scC[u], prC[u], -mC, la, lb);
                asm volatile("" : "+v"(prC[u]));
            }
            lC = row_sum(la, lb);
        }
    };
    auto phase_x = [&](auto &&before, auto &&between) {
        __builtin_amdgcn_s_setprio(0);  // the softmax is always ready to issue: it takes what the MFMA / LDS phases of the other group leave
        before();
        lA = softmax(scA, prA);
        between();
        lB = softmax(scB, prB);
        __builtin_amdgcn_s_setprio(2);
    };
    // O^T of both tiles (every V^T fragment is read once), stores (6 vector stores), partial O^T of the CLS row -> scratch; V is in buffer vb
    auto phase_pv = [&](int vb, bf16 *ob, auto &&step_hook, auto &&between, auto &&ts) {
        unsigned vaddr;
        {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int l15o = ln & 15, go = ln >> 4;
            const int vi = 4 * go + (l15o >> 2);
            vaddr = (unsigned)(uintptr_t)lds + vb * BUF + (vi < 8 ? 2 * vi : 2 * vi - 15) * RS + (l15o & 3) * 8;
        }
        constexpr bf16x4 z4 = {0, 0, 0, 0};
        {
            f32x4 oA[DT], oB[DT];
            // software pipeline in half steps: the V^T fragments of d tiles 0-2 (H0) and 3-5 (H1) of a key-tile pair are separate
            // register sets, and each is re-requested for the next pair as soon as its MFMAs have been issued: six reads are in flight
            // under every group of six MFMAs (issue-wait-compute per whole step left the matrix pipe idle for an LDS round trip per step)
            bf16x4 vf[2 * DT];
            auto issue_half = [&](bf16x4 (&vf)[2 * DT], auto k_c, auto h_c) {
                constexpr int k2 = decltype(k_c)::value, hh = decltype(h_c)::value;
                constexpr bool two = 2 * k2 + 1 < NT;
#pragma unroll
                for (int dt = 3 * hh; dt < 3 * hh + 3; ++dt) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vf[2 * dt]) : "v"(vaddr), "i"(2 * k2 * 16 * RS + dt * 32));
                    if constexpr (two)
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vf[2 * dt + 1]) : "v"(vaddr), "i"((2 * k2 + 1) * 16 * RS + dt * 32));
                }
            };
            auto wait_half = [&](bf16x4 (&vf)[2 * DT], auto h_c, auto n_c) {  // N younger reads may stay in flight
                constexpr int hh = decltype(h_c)::value, N = decltype(n_c)::value;
                asm volatile("s_waitcnt lgkmcnt(%6)"
                             : "+v"(vf[6 * hh]), "+v"(vf[6 * hh + 1]), "+v"(vf[6 * hh + 2]), "+v"(vf[6 * hh + 3]), "+v"(vf[6 * hh + 4]), "+v"(vf[6 * hh + 5])
                             : "n"(N));
            };
            // (the first key step starts from the inline constant 0: no accumulator initialisation, 48 v_mov per wave and pair)
            auto mma_half = [&](const bf16x4 (&vf)[2 * DT], f32x4 (&o)[DT], bf16x4 p0, bf16x4 p1, auto h_c, auto two_c, auto first_c) {
                constexpr int hh = decltype(h_c)::value;
                constexpr bool two = decltype(two_c)::value, first = decltype(first_c)::value;
                const bf16x8 pb = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
#pragma unroll
                for (int dt = 3 * hh; dt < 3 * hh + 3; ++dt) {
                    const bf16x4 lo = vf[2 * dt], hi = two ? vf[2 * dt + 1] : z4;
                    const bf16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if constexpr (first) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v8, pb, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v8, pb, o[dt], 0, 0, 0);
                }
            };
            using H0 = std::integral_constant<int, 0>;
            using H1 = std::integral_constant<int, 1>;
            issue_half(vf, I0{}, H0{});
            issue_half(vf, I0{}, H1{});
            attn_static_for<NKS>([&](auto k_c) {
                constexpr int k2 = decltype(k_c)::value;
                constexpr bool two = 2 * k2 + 1 < NT, last = k2 + 1 == NKS;
                constexpr int n_this = two ? 6 : 3;                                    // reads per half of this step
                constexpr int n_next = last ? 0 : (2 * (k2 + 1) + 1 < NT ? 6 : 3);    // ... of the next step
                using TWO = std::integral_constant<bool, two>;
                using FIRST = std::integral_constant<bool, k2 == 0>;
                const bf16x4 pa1 = two ? prA[two ? 2 * k2 + 1 : 0] : z4, pb1 = two ? prB[two ? 2 * k2 + 1 : 0] : z4;
                wait_half(vf, H0{}, std::integral_constant<int, n_this>{});  // H1 of this step is younger
                mma_half(vf, oA, prA[2 * k2], pa1, H0{}, TWO{}, FIRST{});
                mma_half(vf, oB, prB[2 * k2], pb1, H0{}, TWO{}, FIRST{});
                if constexpr (!last) issue_half(vf, std::integral_constant<int, last ? k2 : k2 + 1>{}, H0{});
                wait_half(vf, H1{}, std::integral_constant<int, n_next>{});  // H0 of the next step is younger
                mma_half(vf, oA, prA[2 * k2], pa1, H1{}, TWO{}, FIRST{});
                mma_half(vf, oB, prB[2 * k2], pb1, H1{}, TWO{}, FIRST{});
                if constexpr (!last) issue_half(vf, std::integral_constant<int, last ? k2 : k2 + 1>{}, H1{});
                step_hook(k_c);
            });
            ts(6);
            store_o(oA, lA, ob, wid);
            store_o(oB, lB, ob, wid + NW);
            ts(7);
        }
        between();  // (the accumulators are free: room for the CLS row's Q fragments of the next pair)
        {
            f32x4 oC[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) oC[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            bf16x4 vf[2 * DT];
            pv_issue(vaddr + (unsigned)wid * (2 * 16 * RS), vf, I0{}, std::true_type{});
            v_wait(vf);
            pv_mma(oC, prC[0], prC[1], vf, std::true_type{});
            if (wid == 0) {
                pv_issue(vaddr, vf, std::integral_constant<int, NKS - 1>{}, std::false_type{});
                v_wait(vf);
                pv_mma(oC, prC[2], z4, vf, std::false_type{});
            }
            if (l15 == 0) {  // query row 16 (NT - 1) + 0 = S - 1: d = 16 dt + 4 g + r
                const unsigned sa = (unsigned)(uintptr_t)lds + SCR + wid * 512 + g * 16;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(sa), "v"(oC[dt]), "i"(dt * 64) : "memory");
                if (g == 0) {
                    const f32x2 ml = {mC, lC};
                    asm volatile("ds_write_b64 %0, %1 offset:384" ::"v"(sa), "v"(ml) : "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };

    // Pair walk.  Head h of a frame is a 176-byte segment of the frame's q / k / v rows: segments of neighbouring heads share 128-byte
    // lines, so a pair fetched on its own moves ~1.7x its bytes.  With workgroup i on XCD i % 8 (round-robin dispatch), XCD x takes the
    // frames x, x + 8, ... and its W workgroups walk them W / heads frames at a time, all heads of a frame together: the shared lines
    // are fetched once into that XCD's L2.  (Grids that do not divide that way walk pair = i, i + G, ...)
    const int G = gridDim.x;
    const bool xcd_walk = G % 8 == 0 && (G / 8) % a.heads == 0;
    const int fpi = xcd_walk ? (G / 8) / a.heads : 0;  // frames per XCD and iteration
    auto pair_at = [&](int k) -> int {  // k-th pair of this workgroup, or npairs
        if (!xcd_walk) {
            const int64_t p = (int64_t)blockIdx.x + (int64_t)k * G;
            return p < npairs ? (int)p : npairs;
        }
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int frame = x + 8 * (fpi * k + j / a.heads);
        return frame < a.batch ? frame * a.heads + j % a.heads : npairs;
    };
    int pair = pair_at(0);
    if (pair >= npairs) return;
    stage_k(pair, 0);
    stage_v(pair, 1);
    load_q01(pair);
    load_qc(pair);
    FA_VMCNT(0);
    const bool ts_on = (a.dbg & 512) && blockIdx.x == 0 && lane == 0;
#define FA_TS(ev)                                                                                  \
    do {                                                                                           \
        if (ts_on && it < 8) g_attn_ts[(wid * 8 + it) * 16 + (ev)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
    __builtin_amdgcn_s_setprio(2);
    FA_BARRIER();  // K(0), V(0) complete in LDS
    bf16 *ob_prev = nullptr;
    if (wid < NW / 2) {
        // ---- group E ---------------------------------------------------------------------------------------------------------
        // vector-memory order per pair: [slot 1] K(p+1) x 6 | [slot 2] Q01(p+1) x 12, V(p+1) x 6, stores(p) x 6, Qc(p+1) x 6
        int kb = 0;  // K(p) in kb, V(p) in kb + 1, K(p+1) in kb + 2, V(p+1) in kb (mod 3)
        for (int it = 0; pair < npairs; pair = pair_at(++it)) {
            const int pn = pair_at(it + 1);
            const bool more = pn < npairs;
            const int vb = kb == 2 ? 0 : kb + 1, kbn = vb == 2 ? 0 : vb + 1;
            FA_TS(0);
            phase_s(kb, true, false, [&](auto) {});
            FA_TS(1);
            FA_BARRIER();
            FA_TS(2);
            if (it > 0) cls_merge(ob_prev);
            const bf16 *ksrc = k_base(more ? pn : pair);
            phase_x(
                [&]() {
                    if (more)
                        for (int k = 0; k < PPW / 2; ++k) stage_piece(ksrc, kbn, k);
                },
                [&]() {
                    if (more)
                        for (int k = PPW / 2; k < PPW; ++k) stage_piece(ksrc, kbn, k);
                });
            FA_TS(3);
            // my pieces of V(p): after them 6 stores, 6 Q loads and the 6 pieces just issued
            if (more) FA_VMCNT(18);
            else FA_VMCNT(0);
            FA_BARRIER();
            FA_TS(4);
            if (more) load_q01(pn);
            const bf16 *vsrc = v_base(more ? pn : pair);
            bf16 *ob = out_base(pair);
            FA_TS(8);
            phase_pv(
                vb, ob,
                [&](auto k_c) {  // one piece of V(p+1) -> the buffer K(p) has left, after each of the first six steps
                    constexpr int k2 = decltype(k_c)::value;
                    if constexpr (k2 < PPW)
                        if (more) stage_piece(vsrc, kb, k2);
                },
                [&]() {
                    if (more) load_qc(pn);
                },
                [&](int ev) { FA_TS(ev); });
            ob_prev = ob;
            FA_TS(5);
            // my pieces of K(p+1): after them 12 Q loads, 6 pieces of V(p+1), 6 stores, 6 Q loads
            if (more) FA_VMCNT(30);
            else FA_VMCNT(0);
            FA_BARRIER();
            kb = kbn;
        }
        FA_BARRIER();  // group L has left its partial of the last pair
        cls_merge(ob_prev);
    } else {
        // ---- group L: one slot behind -----------------------------------------------------------------------------------------
        // vector-memory order per pair: [slot 0] Q01(p) x 12, stores(p-1) x 6, Qc(p) x 6 | [slot 1] K(p+1) x 6 | [slot 2] V(p+1) x 6
        int kb = 0, vb_prev = 0;
        for (int it = 0; pair < npairs; pair = pair_at(++it)) {
            const int pn = pair_at(it + 1);
            const bool more = pn < npairs;
            const int vb = kb == 2 ? 0 : kb + 1, kbn = vb == 2 ? 0 : vb + 1;
            FA_TS(0);
            if (it > 0) {
                FA_TS(8);
                phase_pv(
                    vb_prev, ob_prev,
                    [&](auto k_c) {  // this pair's Q under the previous pair's PV
                        constexpr int k2 = decltype(k_c)::value;
                        if constexpr (k2 == 0) load_q(pair, wid, qlo[0], qhi[0]);
                        if constexpr (k2 == 3) load_q(pair, wid + NW, qlo[1], qhi[1]);
                    },
                    [&]() { load_qc(pair); }, [&](int ev) { FA_TS(ev); });
            }
            FA_TS(1);
            FA_BARRIER();
            FA_TS(2);
            if (it > 0) cls_merge(ob_prev);
            const bf16 *ksrc = k_base(more ? pn : pair);
            phase_s(kb, false, more, [&](auto t_c) {  // one piece of K(p+1) after every other key tile
                constexpr int t = decltype(t_c)::value;
                if constexpr (t % 2 == 0 && t / 2 < PPW)
                    if (more) stage_piece(ksrc, kbn, t / 2);
            });
            FA_TS(3);
            // my pieces of V(p): after them 12 Q loads, 6 stores, 6 Q loads and the 6 pieces just issued
            if (more) FA_VMCNT(30);
            else FA_VMCNT(0);
            FA_BARRIER();
            FA_TS(4);
            const bf16 *vsrc = v_base(more ? pn : pair);
            phase_x(
                [&]() {
                    if (more)
                        for (int k = 0; k < PPW / 2; ++k) stage_piece(vsrc, kb, k);
                },
                [&]() {
                    if (more)
                        for (int k = PPW / 2; k < PPW; ++k) stage_piece(vsrc, kb, k);
                });
            FA_TS(5);
            // my pieces of K(p+1): after them the 6 pieces of V(p+1)
            if (more) FA_VMCNT(6);
            else FA_VMCNT(0);
            FA_BARRIER();
            ob_prev = out_base(pair);
            vb_prev = vb;
            kb = kbn;
        }
        phase_pv(vb_prev, ob_prev, [&](auto) {}, [&]() {}, [&](int) {});
        FA_BARRIER();
        cls_merge(ob_prev);
    }
#undef FA_TS
#undef FA_QUAD
#undef FA_QWAIT2
#undef FA_QWAIT
#undef FA_VMCNT
#undef FA_BARRIER
}

template <int HD, int NT, bool HM = false>
int launch_attn_frame3(const AttnArgs &a, hipStream_t s) {
    constexpr int CH = HD / 8;
    constexpr int NPIECE = (NT * 16 * CH + (12 - CH) + 63) / 64;
    constexpr int smem = 3 * NPIECE * 1024 + 8 * 512;
    static bool attr_set = false;
    static int num_cu = 0;
    if (!attr_set) {
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(attn_frame3_kernel<HD, NT, HM>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        int dev = 0;
        EILEV_HIP_CHECK(hipGetDevice(&dev));
        EILEV_HIP_CHECK(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
        attr_set = true;
    }
    const int npairs = a.batch * a.heads;
    const int ncu = eilev_grid_cus() < num_cu ? eilev_grid_cus() : num_cu;
    const int grid = npairs < ncu ? npairs : ncu;
    hipLaunchKernelGGL((attn_frame3_kernel<HD, NT, HM>), dim3(grid), dim3(512), smem, s, a);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}
