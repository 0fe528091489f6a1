"""Generator of the hand-scheduled tile step of the one-wave-per-SIMD prefill attention kernel (64 query rows per wave,
two 32-row blocks A / B, 64-key tiles): ONE instruction stream per step in which every register is fixed, every
`s_waitcnt lgkmcnt` is counted from a simulated LDS queue, and the softmax of one tile is sliced between the MFMAs of the
neighbouring tiles:

    phase 1 (32 MFMAs)  S(t+1) = K(t+1) Q^T for both blocks (a K fragment feeds two MFMAs)   beside  finish(t): exp2, row sums, bf16 pack
    phase 2 (32 MFMAs)  O += V(t)^T P(t) for both blocks (a V fragment feeds two MFMAs)       beside  start(t+1): row maxima, x = S c - m

Register file (per lane):  v[0:63] score buffer 0 (tuples A.kb0, A.kb1, B.kb0, B.kb1), v[64:127] score buffer 1 (P words
overwrite the scores they came from: word k of a tuple <- scores 2k, 2k+1), v[128:143] fragment ring (4 x 128 bit),
v144.. running statistics / temporaries / LDS addresses;  a[0:127] O (block X, 32-wide head-dim block db: a[64X+16db ..]),
a[128:191] Q fragments (block X, ds: a[128+32X+4ds ..]), a[192:223] staged K / V rows.

`--probe` writes tools/probes/attn_stream_probe.hip: the steady-state step in a timing loop on synthetic LDS contents (no
global traffic, results meaningless) — the go / no-go measurement of the stream's issue rate before the real kernel.
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEL = int(os.environ.get("NVL_PF64_DEL", "0"))     # deletion probes (wrong results): 1 no global stage loads, 2 no LDS stage
#                                                    writes, 4 no decisions, 8 no barrier, 16 no finish VALU, 32 no start VALU

K_ROW = 256
V_ROW = 320
KBUF = 64 * K_ROW      # 16 KiB
VBUF = 64 * V_ROW      # 20 KiB
V_BASE = 2 * KBUF

# ---- register map -----------------------------------------------------------------------------------------------------
def S(buf, X, kb):          # first VGPR of score tuple
    return 64 * buf + 32 * X + 16 * kb
RING = 128                  # 4 fragments x 4 VGPRs
M_RUN = (144, 145)
L_RUN = (146, 147)
MX = ((148, 149), (150, 151))      # [X][kb]
PS = ((152, 153), (154, 155))      # [X][chain]
T = (156, 157, 158, 159)           # temporaries A: T[0:2], B: T[2:4]
KSLOT = 164                 # 8 VGPRs: K fragment LDS byte offsets (ds = 0 .. 7), buffer 0
VLANE = 172                 # V transpose-read lane base (byte offset of V buffer 0 included)
TMPX = (160, 161, 162, 163)


def O(X, db):
    return 64 * X + 16 * db


def Q(X, ds):
    return 128 + 32 * X + 4 * ds


def vr(a, n=1):
    return f"v{a}" if n == 1 else f"v[{a}:{a + n - 1}]"


def ar(a, n=1):
    return f"a{a}" if n == 1 else f"a[{a}:{a + n - 1}]"


class Stream:
    """Instruction list + the LDS queue simulation that turns 'fragment f must have landed' into lgkmcnt(N)."""

    def __init__(self):
        self.ins: list[str] = []
        self.queue: list[str] = []      # tags of LDS operations in flight, oldest first (LDS returns in order)

    def emit(self, s):
        self.ins.append(s)

    def lds(self, s, tag):
        self.ins.append(s)
        self.queue.append(tag)

    def wait_for(self, tag):
        """all operations tagged `tag` have completed"""
        idx = max((i for i, t in enumerate(self.queue) if t == tag), default=None)
        if idx is None:
            return
        n = len(self.queue) - 1 - idx
        self.ins.append(f"s_waitcnt lgkmcnt({n})")
        self.queue = self.queue[idx + 1:]

    def drain(self):
        if self.queue:
            self.ins.append("s_waitcnt lgkmcnt(0)")
            self.queue = []

    def text(self):
        return "\n".join(self.ins)


def read_k(st: Stream, h, kbuf, slot):
    """K fragment h (kb = h >> 3, ds = h & 7) of K buffer kbuf into ring slot"""
    off = kbuf * KBUF + (h >> 3) * 32 * K_ROW
    st.lds(f"ds_read_b128 {vr(RING + 4 * slot, 4)}, {vr(KSLOT + (h & 7))} offset:{off}", f"f{slot}")


def read_v(st: Stream, j, vbuf, slot):
    """V fragment j ((kb, r0) = j >> 2, db = j & 3) of V buffer vbuf into ring slot: two transpose reads"""
    off = vbuf * VBUF + (j >> 2) * 16 * V_ROW + (j & 3) * 64
    st.lds(f"ds_read_b64_tr_b16 {vr(RING + 4 * slot, 2)}, {vr(VLANE)} offset:{off}", f"f{slot}")
    st.lds(f"ds_read_b64_tr_b16 {vr(RING + 4 * slot + 2, 2)}, {vr(VLANE)} offset:{off + 8 * V_ROW}", f"f{slot}")


def finish_chunk(st: Stream, cur, X, c, packed=False):
    """scores 2k, 2k+1 of tuple kb (c = 8 kb + k) -> exp2, row sums, packed word k"""
    if DEL & 16:
        return
    kb, k = c >> 3, c & 7
    base = S(cur, X, kb)
    x0, x1 = base + 2 * k, base + 2 * k + 1
    t0, t1 = T[2 * X], T[2 * X + 1]
    op = "v_mov_b32" if os.environ.get("NVL_PF64_NOEXP") == "1" else "v_exp_f32"     # (probe: what the transcendentals cost)
    st.emit(f"{op} {vr(t0)}, {vr(x0)}")
    st.emit(f"{op} {vr(t1)}, {vr(x1)}")
    if c == 0:
        st.emit(f"v_mov_b32 {vr(PS[X][0])}, {vr(t0)}")
        st.emit(f"v_mov_b32 {vr(PS[X][1])}, {vr(t1)}")
    else:
        st.emit(f"v_add_f32 {vr(PS[X][0])}, {vr(PS[X][0])}, {vr(t0)}")
        st.emit(f"v_add_f32 {vr(PS[X][1])}, {vr(PS[X][1])}, {vr(t1)}")
    st.emit(f"v_cvt_pk_bf16_f32 {vr(base + k)}, {vr(t0)}, {vr(t1)}")


def finish_tail(st: Stream, X):
    st.emit(f"v_add_f32 {vr(PS[X][0])}, {vr(PS[X][0])}, {vr(PS[X][1])}")
    st.emit(f"v_add_f32 {vr(L_RUN[X])}, {vr(L_RUN[X])}, {vr(PS[X][0])}")


def max_part(st: Stream, nxt, X, kb, half):
    if DEL & 32:
        return
    base = S(nxt, X, kb)
    m = MX[X][kb]
    if half == 0:
        st.emit(f"v_max3_f32 {vr(m)}, {vr(base)}, {vr(base + 1)}, {vr(base + 2)}")
        for i in (3, 5, 7):
            st.emit(f"v_max3_f32 {vr(m)}, {vr(m)}, {vr(base + i)}, {vr(base + i + 1)}")
    else:
        for i in (9, 11, 13):
            st.emit(f"v_max3_f32 {vr(m)}, {vr(m)}, {vr(base + i)}, {vr(base + i + 1)}")
        st.emit(f"v_max_f32 {vr(m)}, {vr(m)}, {vr(base + 15)}")


def fma2(st: Stream, nxt, X, e0, scale="s_scale"):
    if DEL & 32:
        return
    for e in (e0, e0 + 1):
        r = S(nxt, X, e >> 4) + (e & 15)
        st.emit(f"v_fma_f32 {vr(r)}, {vr(r)}, {scale}, -{vr(M_RUN[X])}")


def mfma_qk(st: Stream, nxt, X, h, slot):
    kb, ds = h >> 3, h & 7
    d = vr(S(nxt, X, kb), 16)
    c = "0" if ds == 0 else d
    st.emit(f"v_mfma_f32_32x32x16_bf16 {d}, {vr(RING + 4 * slot, 4)}, {ar(Q(X, ds), 4)}, {c}")


def mfma_pv(st: Stream, cur, X, j, slot):
    kbr, db = j >> 2, j & 3
    p = S(cur, X, kbr >> 1) + 4 * (kbr & 1)
    o = ar(O(X, db), 16)
    st.emit(f"v_mfma_f32_32x32x16_bf16 {o}, {vr(RING + 4 * slot, 4)}, {vr(p, 4)}, {o}")


def phase1(st: Stream, cur, nxt, kbuf_next, vbuf_cur, nxt_tile=True):
    """QK^T of the next tile (if any) beside finish of the current one; the last three positions prefetch phase 2's fragments"""
    for h in range(16):
        slot = (h + 3) & 3
        if h + 3 < 16:
            if nxt_tile:
                read_k(st, h + 3, kbuf_next, slot)
        else:
            read_v(st, h + 3 - 16, vbuf_cur, slot)
        finish_chunk(st, cur, 0, h)
        if nxt_tile:
            st.wait_for(f"f{h & 3}")
            mfma_qk(st, nxt, 0, h, h & 3)
        finish_chunk(st, cur, 1, h)
        if nxt_tile:
            mfma_qk(st, nxt, 1, h, h & 3)
    finish_tail(st, 0)
    finish_tail(st, 1)


def phase2(st: Stream, cur, nxt, vbuf_cur, nxt_tile=True):
    for j in range(16):
        if j + 3 < 16:
            read_v(st, j + 3, vbuf_cur, (j + 3) & 3)
        for X in (0, 1):
            if nxt_tile:
                if j < 4:
                    max_part(st, nxt, X, j >> 1, j & 1)
                fma2(st, nxt, X, 2 * j)
            if X == 0:
                st.wait_for(f"f{j & 3}")
            mfma_pv(st, cur, X, j, j & 3)


def steady_step(par, packed=False):
    """the unmasked steady-state step: x(t) in score buffer par, K(t+1) in K buffer par ^ 1, V(t) in V buffer par"""
    st = Stream()
    cur, nxt = par, par ^ 1
    # the first three K fragments of phase 1
    for h in range(3):
        read_k(st, h, par ^ 1, h)
    phase1(st, cur, nxt, par ^ 1, par)
    phase2(st, cur, nxt, par)
    return st



# ---- the real kernel's stream ---------------------------------------------------------------------------------------------
# further fixed registers
KWR, VWR = 173, 174            # LDS write lane offsets of the staged K / V chunk 0 (buffer 0)
LIM = (175, 176)               # per block: last visible key of this lane's row minus the tile's first key (minus 4 hi)
KVOFF = 177                    # 4 VGPRs: global byte offsets of this thread's K chunks 0 .. 3 (V: + 4)
VVOFF = 181
NEGBIG = 185
KMAXV = (186, 187)             # per block: last visible key of this lane's row, minus 4 hi
STG = 192                      # a[192:207] staged K rows, a[208:223] staged V rows
PAGED = False                  # set by core_include(): K / V tiles from the paged cache (a per-tile offset table in LDS)
TABV = (201, 202, 203)          # table address, entry lo / hi
SRD_K, SRD_V = "s[68:71]", "s[72:75]"
SREG = dict(s_tab="s62", s_olo="s63", s_ohi="s64", s_plo="s65", s_phi="s66", s_scale="s40", s_thr="s41", s_t="s42", s_ntw="s43", s_tmask="s44", s_ksoff="s45", s_vsoff="s46", s_ktile="s47",
            s_vtile="s48", s_ktn="s49", s_m0="s[50:51]", s_m1="s[52:53]", s_m2="s[54:55]", s_tmp="s56", s_tmp2="s57")


def stage_load(st: Stream, i):
    """global -> AGPRs, request i of 8: K rows of tile t + 2 (i < 4), V rows of tile t + 1 (rows past the end read as zeros:
    range-checked SRDs); one request per second position of phase 1"""
    if DEL & 1:
        return
    if PAGED:
        if i < 4:
            st.emit(f"buffer_load_dwordx4 {ar(STG + 4 * i, 4)}, {vr(KVOFF + i)}, {SRD_K}, 0 offen")
        else:
            st.emit(f"buffer_load_dwordx4 {ar(STG + 4 * i, 4)}, {vr(VVOFF + i - 4)}, {SRD_V}, 0 offen")
        return
    if i < 4:
        st.emit(f"buffer_load_dwordx4 {ar(STG + 4 * i, 4)}, {vr(KVOFF + i)}, %[ksrd], s_ksoff offen")
    else:
        st.emit(f"buffer_load_dwordx4 {ar(STG + 4 * i, 4)}, {vr(VVOFF + i - 4)}, %[vsrd], s_vsoff offen")
    if i == 7:
        st.emit("s_add_u32 s_ksoff, s_ksoff, s_ktile")
        st.emit("s_add_u32 s_vsoff, s_vsoff, s_vtile")


def paged_lookup(st: Stream):
    """paged cache: request the byte offset of K tile t + 2 from the LDS table (entry i = tile i's offset inside either cache)"""
    st.emit(f"v_mov_b32 {vr(TABV[0])}, s_tab")
    st.lds(f"ds_read_b64 {vr(TABV[1], 2)}, {vr(TABV[0])}", "tb")
    st.emit("s_add_u32 s_tab, s_tab, 8")


def paged_srds(st: Stream):
    """... and turn it into the two descriptors: K tile t + 2 at the new offset, V tile t + 1 at the previous step's"""
    st.wait_for("tb")
    st.emit(f"v_readfirstlane_b32 s_olo, {vr(TABV[1])}")
    st.emit(f"v_readfirstlane_b32 s_ohi, {vr(TABV[2])}")
    st.emit("s_add_u32 s72, %[vb0], s_plo")
    st.emit("s_addc_u32 s73, %[vb1], s_phi")
    st.emit("s_add_u32 s68, %[kb0], s_olo")
    st.emit("s_addc_u32 s69, %[kb1], s_ohi")
    st.emit("s_mov_b32 s_plo, s_olo")
    st.emit("s_mov_b32 s_phi, s_ohi")


def stage_write(st: Stream, i, kbuf, vbuf):
    """AGPRs -> LDS, store i of 8 (one per position in the second half of phase 2: the requests are >= 17 positions old)"""
    if DEL & 2:
        return
    if i == 0:
        st.emit("s_waitcnt vmcnt(0)")
    if i < 4:
        st.lds(f"ds_write_b128 {vr(KWR)}, {ar(STG + 4 * i, 4)} offset:{kbuf * KBUF + i * 16 * K_ROW}", "w")
    else:
        st.lds(f"ds_write_b128 {vr(VWR)}, {ar(STG + 4 * i, 4)} offset:{vbuf * VBUF + (i - 4) * 16 * V_ROW}", "w")   # (VWR holds V_BASE)


def mask_tuple(st: Stream, nxt, X, kb):
    """scores of keys past the lane's causal frontier -> -1e30 (key(kb, r) = kt + 4 hi + kb*32 + (r & 3) + 8 (r >> 2))"""
    base = S(nxt, X, kb)
    pairs = ["s_m0", "s_m1", "s_m2"]
    pend = []
    for r in range(16):
        c = kb * 32 + (r & 3) + 8 * (r >> 2)
        sp = pairs[r % 3]
        st.emit(f"v_cmp_ge_i32_e64 {sp}, {vr(LIM[X])}, {c}")
        pend.append((sp, base + r))
        if len(pend) == 3:
            sp0, reg = pend.pop(0)
            st.emit(f"v_cndmask_b32_e64 {vr(reg)}, {vr(NEGBIG)}, {vr(reg)}, {sp0}")
    st.emit("s_nop 1")
    for sp0, reg in pend:
        st.emit(f"v_cndmask_b32_e64 {vr(reg)}, {vr(NEGBIG)}, {vr(reg)}, {sp0}")


def rescale_block(st: Stream, X):
    """O_X *= alpha (TMPX[0]) through four temporaries"""
    t = (T[0], T[1], T[2], T[3])
    for g in range(16):
        regs = [64 * X + 4 * g + i for i in range(4)]
        for i, r in enumerate(regs):
            st.emit(f"v_accvgpr_read_b32 {vr(t[i])}, {ar(r)}")
        for i in range(4):
            st.emit(f"v_mul_f32 {vr(t[i])}, {vr(t[i])}, {vr(TMPX[0])}")
        for i, r in enumerate(regs):
            st.emit(f"v_accvgpr_write_b32 {ar(r)}, {vr(t[i])}")


MNEW = (188, 189)              # per block: the candidate maximum of the tile being started
DTMP = (190, 191)
DFLAG = ("s[58:59]", "s[60:61]")   # per block: lanes whose maximum grew by more than the threshold


def decide_piece(st: Stream, X, piece):
    """the decision of block X in four pieces (positions 4 .. 7 of phase 2, after the maxima): cross-half maximum, candidate
    maximum, lanes beyond the threshold -> DFLAG. Other instructions of the position separate the pieces (the permlane needs two
    wait states behind the v_mov)."""
    a, b = MNEW[X], DTMP[X]
    if DEL & 4:
        return
    if piece == 0:
        st.emit(f"v_max_f32 {vr(a)}, {vr(MX[X][0])}, {vr(MX[X][1])}")
        st.emit(f"v_mov_b32 {vr(b)}, {vr(a)}")
    elif piece == 1:
        st.emit(f"v_permlane32_swap_b32 {vr(a)}, {vr(b)}")
    elif piece == 2:
        st.emit(f"v_max_f32 {vr(a)}, {vr(a)}, {vr(b)}")
        st.emit(f"v_mul_f32 {vr(a)}, s_scale, {vr(a)}")
        st.emit(f"v_max_f32 {vr(a)}, {vr(a)}, {vr(M_RUN[X])}")          # m_new
    else:
        st.emit(f"v_add_f32 {vr(b)}, s_thr, {vr(M_RUN[X])}")
        st.emit(f"v_cmp_gt_f32_e64 {DFLAG[X]}, {vr(a)}, {vr(b)}")


def decide(st: Stream, nxt, X, tag):
    """after the step's last P.V MFMA: if some row's maximum grew beyond the threshold (DFLAG, rare), bring O, l and the next
    tile's x (taken against the stale maximum) to the new one"""
    a, b = MNEW[X], DTMP[X]
    st.emit(f"s_cmp_lg_u64 {DFLAG[X]}, 0")
    st.emit(f"s_cbranch_scc0 SKIP_{tag}_%=")
    st.emit("s_nop 15")
    st.emit(f"v_sub_f32 {vr(b)}, {vr(M_RUN[X])}, {vr(a)}")            # d = m_old - m_new <= 0
    st.emit(f"v_exp_f32 {vr(TMPX[0])}, {vr(b)}")
    st.emit(f"v_mov_b32 {vr(M_RUN[X])}, {vr(a)}")
    st.emit(f"v_mul_f32 {vr(L_RUN[X])}, {vr(L_RUN[X])}, {vr(TMPX[0])}")
    rescale_block(st, X)
    for kb in (0, 1):
        for r in range(16):
            reg = S(nxt, X, kb) + r
            st.emit(f"v_add_f32 {vr(reg)}, {vr(reg)}, {vr(b)}")
    st.emit("s_nop 3")
    st.emit(f"SKIP_{tag}_%=:")


NF1 = 12        # finish chunks per block done in phase 1; the rest ride in the first positions of phase 2 (P.V j needs chunks <= j)


def phase1_real(st: Stream, cur, nxt, kbuf_next, vbuf_cur, nxt_tile):
    """A gap: block A's finish chunk (+ a staging request every second position), the wait for fragment h; B gap: the read of
    fragment h + 3 and block B's chunk — 6-7 fillers either side"""
    for h in range(16):
        if h < NF1:
            finish_chunk(st, cur, 0, h)
        if PAGED:
            if h == 1:
                paged_srds(st)
            if h >= 2 and h % 2 == 0:
                stage_load(st, h // 2 - 1)
            if h == 15:
                stage_load(st, 7)
        elif h % 2 == 0:
            stage_load(st, h // 2)
        if nxt_tile:
            st.wait_for(f"f{h & 3}")
            mfma_qk(st, nxt, 0, h, h & 3)
        slot = (h + 3) & 3
        if h + 3 < 16:
            if nxt_tile:
                read_k(st, h + 3, kbuf_next, slot)
        else:
            read_v(st, h + 3 - 16, vbuf_cur, slot)
        if h < NF1:
            finish_chunk(st, cur, 1, h)
        if nxt_tile:
            mfma_qk(st, nxt, 1, h, h & 3)


def fma_elems(j):
    """score elements (per block) turned into x = S c - m at position j of phase 2: one each in the first half (beside the
    maxima / the last finish chunks), three each in the second; ascending, so that key block 0's raw scores (read by the
    maxima at positions 0, 1) and key block 1's (positions 2, 3) are never ahead of their readers"""
    return [j] if j < 8 else [8 + 3 * (j - 8) + i for i in range(3)]


def fma1(st: Stream, nxt, X, e):
    if DEL & 32:
        return
    r = S(nxt, X, e >> 4) + (e & 15)
    st.emit(f"v_fma_f32 {vr(r)}, {vr(r)}, s_scale, -{vr(M_RUN[X])}")


def phase2_real(st: Stream, cur, nxt, kbuf_w, vbuf_cur, vbuf_w, nxt_tile, masked):
    for j in range(16):
        for X in (0, 1):
            if X == 1:
                if j + 3 < 16:
                    read_v(st, j + 3, vbuf_cur, (j + 3) & 3)
                if j >= 8:
                    stage_write(st, j - 8, kbuf_w, vbuf_w)
            if nxt_tile:
                if masked and j in (0, 2):
                    mask_tuple(st, nxt, X, j >> 1)
                if j < 4:
                    max_part(st, nxt, X, j >> 1, j & 1)
            if 4 <= j < 4 + 16 - NF1:
                finish_chunk(st, cur, X, NF1 + j - 4)
                if j == 4 + 16 - NF1 - 1:
                    finish_tail(st, X)
            if nxt_tile:
                for e in fma_elems(j):
                    fma1(st, nxt, X, e)
                if 8 <= j < 12:
                    decide_piece(st, X, j - 8)
            if X == 0:
                st.wait_for(f"f{j & 3}")
            mfma_pv(st, cur, X, j, j & 3)


def real_step(par, nxt_tile, masked, tag):
    """one tile step t (t & 1 = par): x(t) in score buffer par, K(t+1) in K buffer par ^ 1, V(t) in V buffer par; stages
    K(t+2) -> K buffer par, V(t+1) -> V buffer par ^ 1; ends with the workgroup barrier"""
    st = Stream()
    cur, nxt = par, par ^ 1
    if PAGED:
        paged_lookup(st)
    if nxt_tile:
        st.emit("s_add_u32 s_ktn, s_ktn, 64")
        if masked:
            for X in (0, 1):
                st.emit(f"v_subrev_u32 {vr(LIM[X])}, s_ktn, {vr(KMAXV[X])}")     # lim = kmax_vis - 4 hi - kt(next)
        for h in range(3):
            read_k(st, h, par ^ 1, h)
    else:
        for j in range(3):
            read_v(st, j, par, j)
    if nxt_tile:
        phase1_real(st, cur, nxt, par ^ 1, par, True)
    else:
        # finish only; the V fragments 0 .. 2 are already requested
        if PAGED:
            paged_srds(st)
        for i in range(8):
            stage_load(st, i)
        for h in range(NF1):
            finish_chunk(st, cur, 0, h)
            finish_chunk(st, cur, 1, h)
    phase2_real(st, cur, nxt, par, par, par ^ 1, nxt_tile, masked)
    if nxt_tile and not (DEL & 4):
        decide(st, nxt, 0, f"{tag}a")
        decide(st, nxt, 1, f"{tag}b")
    st.drain()
    if not (DEL & 8):
        st.emit("s_barrier")
    return st


def first_tile(st: Stream):
    """S(0) = K(0) Q^T (K buffer 0) and its start, not overlapped: mask (always), maxima, m = max c, x = S c - m into score buffer 0"""
    for h in range(3):
        read_k(st, h, 0, h)
    for h in range(16):
        if h + 3 < 16:
            read_k(st, h + 3, 0, (h + 3) & 3)
        st.wait_for(f"f{h & 3}")
        mfma_qk(st, 0, 0, h, h & 3)
        mfma_qk(st, 0, 1, h, h & 3)
    st.emit("s_nop 15")
    st.emit("s_nop 3")
    for X in (0, 1):
        st.emit(f"v_mov_b32 {vr(LIM[X])}, {vr(KMAXV[X])}")        # kt = 0
        for kb in (0, 1):
            mask_tuple(st, 0, X, kb)
        for kb in (0, 1):
            max_part(st, 0, X, kb, 0)
            max_part(st, 0, X, kb, 1)
        a, b = TMPX[1], TMPX[2]
        st.emit(f"v_max_f32 {vr(a)}, {vr(MX[X][0])}, {vr(MX[X][1])}")
        st.emit(f"v_mov_b32 {vr(b)}, {vr(a)}")
        st.emit("s_nop 1")
        st.emit(f"v_permlane32_swap_b32 {vr(a)}, {vr(b)}")
        st.emit("s_nop 0")
        st.emit(f"v_max_f32 {vr(a)}, {vr(a)}, {vr(b)}")
        st.emit(f"v_mul_f32 {vr(M_RUN[X])}, s_scale, {vr(a)}")
        for e in range(0, 32, 2):
            fma2(st, 0, X, e)


def kernel_stream():
    """the whole per-wave main loop: prologue (Q, O, first tile), the step variants, the control flow between them"""
    st = Stream()
    e = st.emit
    # ---- inputs into the fixed registers -----------------------------------------------------------------------------
    # (hipcc pads nothing in FRONT of an asm statement: its last VALU write of an input and our first v_readlane of it need
    #  a wait state between them — found as a wrong softmax scale in the paged instantiation only)
    e("s_nop 4")
    for i, name in enumerate(("s_scale", "s_thr", "s_ntw", "s_tmask", "s_ktile", "s_vtile", "s_ksoff", "s_vsoff")):
        e(f"v_readlane_b32 {name}, %[pk], {i}")
    e(f"v_mov_b32 {vr(VLANE)}, %[vl]")
    e(f"v_mov_b32 {vr(KWR)}, %[kwr]")
    e(f"v_mov_b32 {vr(VWR)}, %[vwr]")
    e(f"v_mov_b32 {vr(KMAXV[0])}, %[kma]")
    e(f"v_mov_b32 {vr(KMAXV[1])}, %[kmb]")
    e(f"v_mov_b32 {vr(NEGBIG)}, 0xf149f2ca")           # -1.0e30f
    for ds in range(8):
        e(f"v_xor_b32 {vr(KSLOT + ds)}, {ds * 32}, %[kk]")
        e(f"v_add_u32 {vr(KSLOT + ds)}, {vr(KSLOT + ds)}, %[rb]")
    # chunk n of a staged tile: + n * 16 rows (s_tmp = 16 rows of K in bytes = tile stride / 4)
    e("s_lshr_b32 s_tmp, s_ktile, 2")
    e("s_lshr_b32 s_tmp2, s_vtile, 2")
    e(f"v_mov_b32 {vr(KVOFF)}, %[ko]")
    e(f"v_mov_b32 {vr(VVOFF)}, %[vo]")
    for n in range(1, 4):
        e(f"v_add_u32 {vr(KVOFF + n)}, s_tmp, {vr(KVOFF + n - 1)}")
        e(f"v_add_u32 {vr(VVOFF + n)}, s_tmp2, {vr(VVOFF + n - 1)}")
    # Q fragments straight into the accumulator file; O = 0, l = 0
    for X, ptr in ((0, "%[qa]"), (1, "%[qb]")):
        for ds in range(8):
            e(f"global_load_dwordx4 {ar(Q(X, ds), 4)}, {ptr}, off offset:{ds * 32}")
    for r in range(128):
        e(f"v_accvgpr_write_b32 {ar(r)}, 0")
    e(f"v_mov_b32 {vr(L_RUN[0])}, 0")
    e(f"v_mov_b32 {vr(L_RUN[1])}, 0")
    e("s_mov_b32 s_t, 0")
    e("s_mov_b32 s_ktn, 0")
    if PAGED:
        # descriptors of one cache tile each (64 rows x 256 B, range-checked); the table pointer starts at entry 1 = V(1)'s
        # offset for step 0 (the previous step's K offset ever after)
        e("v_readlane_b32 s_tab, %[pk], 8")
        for base in (70, 74):
            e(f"s_mov_b32 s{base}, 0x4000")
            e(f"s_mov_b32 s{base + 1}, 0x00020000")
        paged_lookup(st)
        st.wait_for("tb")
        e(f"v_readfirstlane_b32 s_plo, {vr(TABV[1])}")
        e(f"v_readfirstlane_b32 s_phi, {vr(TABV[2])}")
    e("s_waitcnt vmcnt(0)")
    first_tile(st)
    st.drain()
    e("s_barrier")          # every wave has read K(0) before the first step overwrites its buffer
    # ---- the loop: even / odd tile index, masked / plain start of the next tile ------------------------------------------
    for par in (0, 1):
        e(f"LOOP{par}_%=:")
        e("s_add_u32 s_tmp, s_t, 1")
        e("s_cmp_ge_u32 s_tmp, s_ntw")
        e(f"s_cbranch_scc1 LAST{par}_%=")
        e("s_cmp_ge_u32 s_tmp, s_tmask")
        e(f"s_cbranch_scc1 MASK{par}_%=")
        for l in real_step(par, True, False, f"p{par}").ins:
            e(l)
        e("s_add_u32 s_t, s_t, 1")
        e(f"s_branch LOOP{par ^ 1}_%=")
        e(f"MASK{par}_%=:")
        for l in real_step(par, True, True, f"m{par}").ins:
            e(l)
        e("s_add_u32 s_t, s_t, 1")
        e(f"s_branch LOOP{par ^ 1}_%=")
    for par in (0, 1):
        e(f"LAST{par}_%=:")
        for l in real_step(par, False, False, f"l{par}").ins:
            e(l)
        e("s_branch END_%=")
    e("END_%=:")
    e("s_nop 15")
    e("s_nop 3")
    e(f"v_mov_b32 %[ma], {vr(M_RUN[0])}")
    e(f"v_mov_b32 %[mb], {vr(M_RUN[1])}")
    e(f"v_mov_b32 %[la], {vr(L_RUN[0])}")
    e(f"v_mov_b32 %[lb], {vr(L_RUN[1])}")
    return st


def core_text(paged):
    global PAGED
    PAGED = paged
    st = kernel_stream()
    PAGED = False
    lines = []
    import re
    pat = re.compile(r"\b(" + "|".join(sorted(SREG, key=len, reverse=True)) + r")\b")
    for l in st.ins:
        lines.append(pat.sub(lambda m: SREG[m.group(1)], l))
    body = "\n".join(f'    "{l}\\n"' for l in lines)
    return len(lines), body.replace(chr(10), " " + chr(92) + chr(10))


def core_include():
    n0, packed = core_text(False)
    n1, paged = core_text(True)
    clob = ", ".join([f'"v{i}"' for i in range(0, 204)] + [f'"a{i}"' for i in range(128, 224)] + [f'"s{i}"' for i in range(40, 76)])
    return f"""// GENERATED by tools/gen_prefill_asm.py — do not edit; the generator's docstring describes the schedule.
// Packed K / V: {n0} instructions, paged cache: {n1} (two plain steps, two masked steps, two last steps, prologue, each).
#define NVL_PF64_CORE_ASM \\
{packed}
#define NVL_PF64_CORE_ASM_PAGED \\
{paged}
#define NVL_PF64_CORE_CLOBBERS {clob}
"""

# ---- the probe --------------------------------------------------------------------------------------------------------
PROBE = r'''// GENERATED by tools/gen_prefill_asm.py --probe — do not edit.
// The steady-state tile step of the planned one-wave-per-SIMD prefill kernel as a timing loop: fixed registers, counted
// lgkmcnt, 64 MFMAs + 48 fragment reads + the softmax slices of two 32-row blocks, on synthetic LDS contents (results are
// meaningless). Prints us per step-iteration, shader cycles per step (s_memtime), and the TFLOP/s the stream would carry
// (64 MFMAs x 32,768 FLOP per wave-step, 1024 waves). Build + run:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/attn_stream_probe.hip -o /tmp/attnstream && /tmp/attnstream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CLOBBERS %(clobbers)s

template <int VAR>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* cyc, float* sink, int iters, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // synthetic K / V tiles: small bf16 values
  for (int i = threadIdx.x; i < (2 * %(kbuf)d + 2 * %(vbuf)d) / 4; i += 256) {
    unsigned int h = (i * 2654435761u) >> 7;
    reinterpret_cast<unsigned int*>(smem)[i] = 0x3c003c00u ^ ((h & 0x3f) << 16) ^ (h & 0x803f);   // ~ +-0.008 .. 0.016
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, qcol = lane & 31, hi = lane >> 5, i16 = lane & 15;
  const int kk = ((hi ^ (qcol & 15)) << 4), rowb = qcol * %(krow)d;
  const int vlane = %(vbase)d + (4 * hi + (i16 >> 2)) * %(vrow)d + (16 * ((lane >> 4) & 1) + (i16 & 3) * 4) * 2;
  unsigned long long t0 = 0, t1 = 0;
  float outv;
  asm volatile(
      // LDS addresses, zeroed state
      "v_mov_b32 v%(vlane)d, %%[vl]\n"
%(kslots)s
      "s_mov_b32 s_scale, %%[sc]\n"
%(init)s
      "s_waitcnt lgkmcnt(0)\n"
      "s_memtime %%[t0]\n"
      "s_waitcnt lgkmcnt(0)\n"
      "s_mov_b32 s_cnt, %%[it]\n"
      "LOOP_%%=:\n"
%(body)s
      "s_sub_u32 s_cnt, s_cnt, 1\n"
      "s_cmp_lg_u32 s_cnt, 0\n"
      "s_cbranch_scc1 LOOP_%%=\n"
      "s_nop 15\n"
      "s_memtime %%[t1]\n"
      "s_waitcnt lgkmcnt(0)\n"
      "v_accvgpr_read_b32 %%[o], a0\n"
      : [t0] "=&s"(t0), [t1] "=&s"(t1), [o] "=&v"(outv)
      : [vl] "v"(vlane), [kk] "v"(kk), [rb] "v"(rowb), [sc] "s"(scale), [it] "s"(iters)
      : "memory", "scc", "vcc", CLOBBERS);
  sink[blockIdx.x * 256 + threadIdx.x] = outv;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  unsigned long long* cyc;
  float* sink;
  hipMalloc(&cyc, 8);
  hipMalloc(&sink, 4 * 65536);
  const int lds = 2 * %(kbuf)d + 2 * %(vbuf)d;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 3; ++rep) {
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), lds, 0, cyc, sink, iters, 0.1275f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double us = ms * 1e3 / iters;
    const double tf = 1024.0 * 64 * 32768 / (us * 1e-6) / 1e12;
    printf("{\"variant\": \"%(name)s\", \"us_per_step\": %%.4f, \"cycles_per_step\": %%.1f, \"clock_GHz\": %%.3f, \"TFLOPs_equivalent\": %%.1f, \"err\": \"%%s\"}\n",
           us, (double)c / iters, (double)c / iters / (us * 1e3), tf, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
'''


def probe_source(name="full"):
    b = Stream()
    s0 = steady_step(0)
    s1 = steady_step(1)
    body_lines = s0.ins + ["s_waitcnt lgkmcnt(0)", "s_barrier"] + s1.ins + ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    # symbolic SGPR names -> fixed registers
    sub = {"s_scale": "s40", "s_cnt": "s41"}
    def fix(line):
        for k, v in sub.items():
            line = line.replace(k, v)
        return line
    body = "\n".join(f'      "{fix(l)}\\n"' for l in body_lines)
    kslots = "\n".join(
        f'      "v_xor_b32 v{KSLOT + ds}, {ds * 32}, %[kk]\\n"\n      "v_add_u32 v{KSLOT + ds}, v{KSLOT + ds}, %[rb]\\n"' for ds in range(8))
    init = []
    for r in range(0, 128):
        init.append(f'      "v_mov_b32 v{r}, 0\\n"')
    for r in (144, 145, 146, 147):
        init.append(f'      "v_mov_b32 v{r}, 0\\n"')
    for r in range(0, 128):
        init.append(f'      "v_accvgpr_write_b32 a{r}, 0\\n"')
    init.append('      "v_mov_b32 v160, 0x3c003c00\\n"')
    for r in range(128, 192):
        init.append(f'      "v_accvgpr_write_b32 a{r}, v160\\n"')
    clob = ", ".join([f'"v{i}"' for i in range(0, 176)] + [f'"a{i}"' for i in range(0, 224)] + ['"s40"', '"s41"'])
    src = PROBE % dict(clobbers=clob, kbuf=KBUF, vbuf=VBUF, krow=K_ROW, vrow=V_ROW, vbase=V_BASE, vlane=VLANE,
                       kslots=fix(kslots), init="\n".join(init), body=body, name=name)
    return fix(src), len(body_lines)


if __name__ == "__main__":
    if "--core" in sys.argv or len(sys.argv) == 1:
        out = os.path.join(ROOT, "nano_vllm_amd", "csrc", "attn_prefill64_core.inc")
        text = core_include()
        if "--check" in sys.argv:
            sys.exit(0 if open(out).read() == text else 1)
        with open(out, "w") as fh:
            fh.write(text)
        print(out, text.count("\\n"), "instructions")
    if "--probe" in sys.argv:
        src, n = probe_source()
        out = os.path.join(ROOT, "tools", "probes", os.environ.get("NVL_PF64_PROBE_NAME", "attn_stream_probe") + ".hip")
        with open(out, "w") as fh:
            fh.write(src)
        print(out, n, "instructions per two steps")
