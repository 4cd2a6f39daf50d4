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
address of the wave's first A piece in buffer 0 (its W pieces: + 32768; buffer 1: + 65536); %28 form of the loop (0 full tile, 1 half tile, 2 half
tile without W pieces: see main()).

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


def dma_pieces(na=8, nw=8, kind="lds"):
    """[(m0 setup, load)] for the wave's pieces of the step at byte offset S_KOFF.  kind (probe variants, results are garbage): "vgpr" =
    the same loads into scratch VGPRs instead of LDS, "nop" = an s_nop in place of the load (issue cost of the LDS-DMA by difference)"""
    out = []
    for i in range(na + nw):
        a = i < na
        ii = i if a else i - na
        m0 = f"s_add_u32 m0, {S_DA if a else S_DW}, {ii * 1024}"
        off, rs = (DA[i] if a else DW[i - na]), ("%22" if a else "%23")
        if kind == "lds4":  # four pieces per M0 value: the instruction offset (added to BOTH addresses) walks the 4 KiB; the prologue
            # subtracted it from the pieces' global offsets
            m0 = f"s_add_u32 m0, {S_DA if a else S_DW}, {(ii // 4) * 4096}" if ii % 4 == 0 else None
            out.append((m0, f"buffer_load_dwordx4 v{off}, {rs}, {S_KOFF} offen offset:{(ii % 4) * 1024} lds"))
        elif kind == "lds":
            out.append((m0, f"buffer_load_dwordx4 v{off}, {rs}, {S_KOFF} offen lds"))
        elif kind == "vgpr":
            out.append((m0, f"buffer_load_dwordx4 v[{152 + 4 * (i % 4)}:{155 + 4 * (i % 4)}], v{off}, {rs}, {S_KOFF} offen"))
        else:
            out.append((m0, "s_nop 0"))
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
    if n != 64:  # half tiles (tn = 2: 32 MFMAs per step): the same schedule on the shorter stream
        v = dict(v, **v.get("half", {}))
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
        for q, (m0set, ld) in enumerate(dma_pieces(na, nw, v.get("dma_kind", "lds"))):
            at = clamp(max(b1 + 1, v["d_at"] + (q * v["d_num"]) // v["d_den"]))
            if at <= h:
                issued_before_h += 1
            if m0set is None:
                post[at].append(ld)
            elif not any(x.startswith(("buffer_load", "s_nop")) for x in post[at]):
                pre[at].append(m0set)
                post[at].append(ld)
            else:  # further pieces in one slot: M0 write, one wait state (SALU write of M0 -> LDS-DMA), load
                post[at] += [m0set, "s_nop 0", ld]
    if nxt:
        # handover: the h-slot's own pieces are issued first (they precede this in the slot), everything older has landed
        post[h].append(f"s_waitcnt vmcnt({issued_before_h if v.get('dma_kind', 'lds') != 'nop' else 0})")
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
        L += [f"s_add_u32 {S_KOFF}, {S_KOFF}, {0 if v.get('freeze_k') else 128}", f"s_xor_b32 {S_DA}, {S_DA}, 0x10000", f"s_xor_b32 {S_DW}, {S_DW}, 0x10000"]
    if nxt:
        L.append("s_waitcnt lgkmcnt(0)")
    return L


def loop_text(v, tm=4, tn=4, na=8, nw=8):
    L = []
    share = v.get("dma_kind", "lds") == "lds4"
    # ---- prologue: per-piece DMA offsets, per-k-slice read addresses, counters
    for i in range(8):
        L.append(f"s_mul_i32 {S_T}, %24, {i}")
        if share and i % 4:
            L.append(f"s_sub_u32 {S_T}, {S_T}, {(i % 4) * 1024}")
        L.append(f"v_add_u32 v{DA[i]}, {S_T}, {'%18' if i % 2 == 0 else '%19'}")
    for i in range(8):
        L.append(f"s_mul_i32 {S_T}, %25, {i}")
        if share and i % 4:
            L.append(f"s_sub_u32 {S_T}, {S_T}, {(i % 4) * 1024}")
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


CLOBBER = ['"memory"', '"scc"', '"vcc"'] + [f'"s{n}"' for n in range(84, 89)] + [f'"v{n}"' for n in range(168)]  # (v152..167: probe variants)

# schedule knobs (see body): positions are MFMA indices 0..63 of the K-step
def K(r1_at=0, r1=(1, 1), b1_at=31, d_at=32, d=(1, 1), h_at=47, h_at_tail=39, r0=(1, 1)):
    return dict(r1_at=r1_at, r1_num=r1[0], r1_den=r1[1], b1_at=b1_at, d_at=d_at, d_num=d[0], d_den=d[1], h_at=h_at, h_at_tail=h_at_tail,
                r0_num=r0[0], r0_den=r0[1])


HALF = dict(r1_at=0, r1_num=1, r1_den=1, b1_at=12, d_at=13, d_num=1, d_den=1, h_at=19, h_at_tail=15, r0_num=1, r0_den=1)
VARIANTS = {
    # measured (profiles/r03_a4_schedule.md): DMA pieces bunched one per MFMA (the first schedule) 1266-1300 TFLOP/s main loop on fc1; one
    # per two MFMAs 1386; one per three 1398-1450; 2 per 5: 1397; one per four 1381.  What the pieces cost is their ISSUE (12 % of the loop:
    # the same loads into VGPRs cost the same, loads that always hit the cache cost the same, no loads at all: 1642), not M0 traffic
    # (four pieces per M0 write: no change) and not memory latency.
    "V0": dict(K(b1_at=17, d_at=18, d=(3, 1), h_at=43), half=HALF),
    "V1": dict(K(b1_at=17, d_at=18, d=(3, 1), h_at=43), half=HALF),   # (experiment slot)
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
        # ONE asm statement holds the three forms of the loop, selected by operand %28 (one statement = one definition of the accumulators:
        # three statements made hipcc merge their 16 x 16 output registers through 256 v_accvgpr_read / write per tile):
        #   0  full tile: 128 x 128 per wave
        #   1  half tile (N % 256 = 128: only columns 0..127 of the tile exist): every wave owns 128 rows x 64 columns (operands i*4 + j, j < 2)
        #   2  half tile, a wave whose W rows (128..255 of the tile) do not exist: stages no W pieces
        L = ["s_cmp_eq_u32 %28, 0", "s_cbranch_scc0 7f"] + loop_text(v) + ["s_branch 9f", "7:", "s_cmp_eq_u32 %28, 1", "s_cbranch_scc0 8f"]
        L += loop_text(v, tn=2) + ["s_branch 9f", "8:"] + loop_text(v, tn=2, nw=0) + ["9:"]
        print(emit(f"A4_LOOP_{vn}", L))
        print()
    print("#define A4_LOOP_CLOBBERS " + ", ".join(CLOBBER))


if __name__ == "__main__":
    main()
