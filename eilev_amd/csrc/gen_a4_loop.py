#!/usr/bin/env python
"""Generates gemm_a4_loop.inc: the hand-scheduled K loop of gemm_a4_kernel (gemm_a4.h) as inline-asm text.

One wave per SIMD, wave tile 128 x 128 (TM = TN = 4 blocks of 32 x 32, all 256 AGPRs are accumulators), K-step 64 = four
k-slices of 16 (v_mfma_f32_32x32x16_bf16: 64 MFMAs per wave and K-step = 2048 matrix-pipe cycles).  hipcc cannot allocate this
geometry (with no spare AGPR every VGPR overflow is a scratch spill), so the loop is written out here with fixed
registers and the C++ kernel only passes operands in and takes the accumulators out.

Register plan (VGPRs clobbered by the asm):
  v[0:63]    fragment set S0 = k-slices 0, 1:  A(i, kk) at (kk*4 + i)*4, W(j, kk) at 32 + (kk*4 + j)*4      (i, j = 0..3)
  v[64:127]  fragment set S1 = k-slices 2, 3
  v[128:131] LDS read addresses of this lane's A row for k-slice 0..3 (the chunk swizzle makes them an XOR of bits 5-6), v[132:135] W
  v[136:143] LDS-DMA byte offsets of the wave's 8 A pieces (8 rows x 128 B each), v[144:151] of its 8 W pieces

Pipeline of K-step s (LDS buffer b = s & 1), the point being that buffer b is free for step s + 2 after a QUARTER of step s:
  phase A  MFMAs of k-slices 0, 1 (S0, read during step s - 1) | the 16 ds_read_b128 of S1 from buffer b between them
           s_waitcnt lgkmcnt(0); s_barrier            -> every wave has read all of buffer b
  phase B  MFMAs of k-slices 2, 3 (S1) | the wave's 16 LDS-DMA pieces of step s + 2 into buffer b, then s_waitcnt vmcnt(16)
           (step s + 1 has landed: issued a whole K-step ago), s_barrier, the 16 ds_read_b128 of S0 of step s + 1 from buffer b ^ 1
so a DMA has ~1.2 K-steps (~2500 shader clocks) between issue and first use, with two LDS buffers: the registers are the third stage.

Operands of the asm statement (gemm_a4.h): %0..%15 accumulators acc[i][j] ("=a": written only — the first k-slice of a tile takes C = 0; index i*4 + j); %16 / %17 LDS read address of the lane's
A / W row (buffer 0, k-slice 0); %18 / %19 DMA offset of the lane in an even / odd A piece, %20 / %21 same for W; %22 / %23 buffer
descriptors of the A / W tile; %24 / %25 bytes between consecutive pieces (8 rows) of A / W; %26 number of K-steps (>= 3); %27 LDS byte
address of the wave's first A piece in buffer 0 (its W pieces: + 32768; buffer 1: + 65536).

    python gen_a4_loop.py [> gemm_a4_loop.inc]     knobs: see VARIANTS
"""
import sys

S_KOFF, S_DA, S_DW, S_CNT, S_T = "s84", "s85", "s86", "s87", "s88"
RA = [128 + k for k in range(4)]
RW = [132 + k for k in range(4)]
DA = [136 + i for i in range(8)]
DW = [144 + i for i in range(8)]


def frag_a(i, k16):
    return (k16 >> 1) * 64 + ((k16 & 1) * 4 + i) * 4


def frag_w(j, k16):
    return (k16 >> 1) * 64 + 32 + ((k16 & 1) * 4 + j) * 4


def mfma(i, j, k16, zero_c=False):
    a, w = frag_a(i, k16), frag_w(j, k16)
    return f"v_mfma_f32_32x32x16_bf16 %{i * 4 + j}, v[{w}:{w + 3}], v[{a}:{a + 3}], {'0' if zero_c else '%' + str(i * 4 + j)}"


def reads(k16s, tm=4, tn=4):
    out = []
    for k in k16s:
        for i in range(tm):
            a = frag_a(i, k)
            out.append(f"ds_read_b128 v[{a}:{a + 3}], v{RA[k]} offset:{i * 4096}")
        for j in range(tn):
            w = frag_w(j, k)
            out.append(f"ds_read_b128 v[{w}:{w + 3}], v{RW[k]} offset:{j * 4096}")
    return out


def group(k16, tm=4, tn=4, zero_c=False):
    return [mfma(i, j, k16, zero_c) for i in range(tm) for j in range(tn)]


def dma_pieces(na=8, nw=8):
    """[(m0 setup, load)] for the wave's pieces of the step at byte offset S_KOFF"""
    out = []
    for i in range(na):
        out.append((f"s_add_u32 m0, {S_DA}, {i * 1024}", f"buffer_load_dwordx4 v{DA[i]}, %22, {S_KOFF} offen lds"))
    for i in range(nw):
        out.append((f"s_add_u32 m0, {S_DW}, {i * 1024}", f"buffer_load_dwordx4 v{DW[i]}, %23, {S_KOFF} offen lds"))
    return out


def body(dma, nxt, v, tm=4, tn=4, na=8, nw=8, first=False):
    """One K-step as ONE stream of 64 MFMAs (k-slices 0..3) with the other instructions placed by MFMA index (v: schedule knobs):
         r1_at + q * r1_num // r1_den   the 16 reads of S1 (k-slices 2, 3 of THIS step, this buffer)
         b1_at                          s_waitcnt lgkmcnt(0) + s_barrier: every wave has read all of this buffer (only when dma)
         d_at + q * d_num // d_den      the 16 LDS-DMA pieces of step s + 2 into this buffer (>= b1_at)
         h_at                           s_waitcnt vmcnt(pieces issued so far) — step s + 1 has landed —, s_barrier, flip the read addresses
         h_at + 1 + q * r0_num // r0_den  the 16 reads of S0 (k-slices 0, 1) of step s + 1
    An item placed "at m" follows MFMA m.  first: the tile's first step — its first 16 MFMAs take C = 0 (the accumulators are outputs
    only: no zero fill, nothing live across the tile boundary).  Returns a list of instructions."""
    mm = group(0, tm, tn, zero_c=first) + group(1, tm, tn) + group(2, tm, tn) + group(3, tm, tn)
    n = len(mm)
    pre = [[] for _ in mm]   # before MFMA m (M0 setup: a SALU write of M0 wants an instruction between it and its user)
    post = [[] for _ in mm]  # after MFMA m
    clamp = lambda x: max(0, min(n - 1, x))
    half = n // 2            # S1 must be complete before the first MFMA of k-slice 2
    for q, r in enumerate(reads([2, 3], tm, tn)):
        post[clamp(min(v["r1_at"] + (q * v["r1_num"]) // v["r1_den"], half - 2))].append(r)
    b1 = clamp(min(v["b1_at"], half - 1))
    post[b1].append("s_waitcnt lgkmcnt(0)")
    if dma:
        post[b1].append("s_barrier")
    h = clamp(v["h_at"] if dma else v["h_at_tail"])
    issued_before_h = 0
    if dma:
        for q, (m0set, ld) in enumerate(dma_pieces(na, nw)):
            at = clamp(max(b1 + 1, v["d_at"] + (q * v["d_num"]) // v["d_den"]))
            if at <= h:
                issued_before_h += 1
            if not any(x.startswith("buffer_load") for x in post[at]):
                pre[at].append(m0set)
                post[at].append(ld)
            else:  # further pieces in one slot: M0 write, one wait state (SALU write of M0 -> LDS-DMA), load
                post[at] += [m0set, "s_nop 0", ld]
    if nxt:
        # handover: the h-slot's own pieces are issued first (they precede this in the slot), everything older has landed
        post[h].append(f"s_waitcnt vmcnt({issued_before_h})")
        post[h].append("s_barrier")
        for r in RA + RW:
            post[h].append(f"v_xor_b32 v{r}, 0x10000, v{r}")
        for q, r in enumerate(reads([0, 1], tm, tn)):
            post[clamp(h + 1 + (q * v["r0_num"]) // v["r0_den"])].append(r)
    L = []
    for m, a, b in zip(mm, pre, post):
        L += a
        L.append(m)
        L += b
    if dma:
        L += [f"s_add_u32 {S_KOFF}, {S_KOFF}, 128", f"s_xor_b32 {S_DA}, {S_DA}, 0x10000", f"s_xor_b32 {S_DW}, {S_DW}, 0x10000"]
    if nxt:
        L.append("s_waitcnt lgkmcnt(0)")
    return L


def loop_text(v, tm=4, tn=4, na=8, nw=8):
    L = []
    # ---- prologue: per-piece DMA offsets, per-k-slice read addresses, counters
    for i in range(8):
        L.append(f"s_mul_i32 {S_T}, %24, {i}")
        L.append(f"v_add_u32 v{DA[i]}, {S_T}, {'%18' if i % 2 == 0 else '%19'}")
    for i in range(8):
        L.append(f"s_mul_i32 {S_T}, %25, {i}")
        L.append(f"v_add_u32 v{DW[i]}, {S_T}, {'%20' if i % 2 == 0 else '%21'}")
    for k in range(4):
        L.append(f"v_xor_b32 v{RA[k]}, {k << 5}, %16")
        L.append(f"v_xor_b32 v{RW[k]}, {k << 5}, %17")
    L += [f"s_mov_b32 {S_KOFF}, 256", f"s_mov_b32 {S_DA}, %27", f"s_add_u32 {S_DW}, %27, 0x8000", f"s_sub_u32 {S_CNT}, %26, 3",
          "s_waitcnt vmcnt(0)", "s_barrier"]
    L += reads([0, 1], tm, tn)
    L.append("s_waitcnt lgkmcnt(0)")
    # ---- step 0 (C = 0 on the first k-slice), then the steady state: steps 1 .. ns - 3; all of them stage step s + 2
    L += body(True, True, v, tm, tn, na, nw, first=True)
    L += [f"s_cmp_eq_u32 {S_CNT}, 0", "s_cbranch_scc1 2f"]
    L.append("1:")
    L += body(True, True, v, tm, tn, na, nw)
    L += [f"s_sub_u32 {S_CNT}, {S_CNT}, 1", f"s_cmp_lg_u32 {S_CNT}, 0", "s_cbranch_scc1 1b"]
    L.append("2:")
    # ---- the last two steps stage nothing
    L += body(False, True, v, tm, tn, na, nw)
    L += body(False, False, v, tm, tn, na, nw)
    return L


CLOBBER = ['"memory"', '"scc"', '"vcc"'] + [f'"s{n}"' for n in range(84, 89)] + [f'"v{n}"' for n in range(152)]

# schedule knobs (see body): positions are MFMA indices 0..63 of the K-step
def K(r1_at=0, r1=(1, 1), b1_at=31, d_at=32, d=(1, 1), h_at=47, h_at_tail=39, r0=(1, 1)):
    return dict(r1_at=r1_at, r1_num=r1[0], r1_den=r1[1], b1_at=b1_at, d_at=d_at, d_num=d[0], d_den=d[1], h_at=h_at, h_at_tail=h_at_tail,
                r0_num=r0[0], r0_den=r0[1])


VARIANTS = {
    "V0": K(),                                                   # the first measured schedule: DMA one per MFMA in 32..47, reads of the next step one per MFMA in 48..63
    "V1": K(b1_at=19, d_at=20, d=(2, 1), h_at=47, r0=(1, 1)),   # buffer freed after MFMA 19, DMA one per two MFMAs in 20..50
    "V2": K(b1_at=19, d_at=20, d=(3, 1), h_at=43, r0=(1, 1)),   # DMA one per three MFMAs in 20..63, hand-over after MFMA 43 (8 pieces issued)
    "V3": K(b1_at=23, d_at=24, d=(5, 2), h_at=44, r0=(1, 1)),   # DMA 2 per 5 MFMAs in 24..61
}


def emit(name, lines):
    out = [f"#define {name} \\"]
    for ins in lines:
        out.append(f'    "{ins}\\n\\t" \\')
    out.append('    ""')
    return "\n".join(out)


def main():
    print("// Generated by gen_a4_loop.py — do not edit.  See that file for the register plan and the pipeline.")
    for vn, v in VARIANTS.items():
        print(emit(f"A4_LOOP_{vn}", loop_text(v)))
        print()
    print("#define A4_LOOP_CLOBBERS " + ", ".join(CLOBBER))


if __name__ == "__main__":
    main()
