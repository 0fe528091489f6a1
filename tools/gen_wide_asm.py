#!/usr/bin/env python3
"""Generates nano_vllm_amd/csrc/gemm_wide_core.inc: the hand-scheduled K loop of the deep-K decode GEMM's consumer waves
(csrc/gemm_wide.hip, `linear_wide_asm_kernel`) as ONE inline-asm statement per row-tile count.

Why generated: the loop is RING unrolled k steps of straight-line code, ~145 instructions each, whose register names,
LDS offsets and wait counts are all functions of (step, fragment slot); writing that by hand once is possible, changing
the schedule is not. This script IS the schedule; the .inc file is committed so that the build needs no Python step.

    python tools/gen_wide_asm.py            # rewrite the .inc
    python tools/gen_wide_asm.py --check    # exit 1 when the committed .inc is stale (tests/test_abi.py)

What the consumer wave does (one wave per SIMD, NT = 2 sixteen-column tiles x MT row tiles, k step = 64 columns):
  * accumulators: MT x 2 x f32x4 in AGPRs, owned by the statement ("+a" operands: hipcc allocates, never touches);
  * W fragments stream HBM -> VGPR through a RING-deep register ring with non-temporal saddr loads, 4 per step, issued
    between the MFMAs; the statement counts them itself: `s_waitcnt vmcnt((RING - 2) * 4)` at the top of a step says
    "the set of THIS step has landed", nothing is ever drained;
  * x fragments come from the LDS stage the loader wave filled (4 stages, XOR-swizzled 16-byte slots: conflict-free
    ds_read_b128), through an 8-entry register ring read D = 7 fragments ahead of their MFMAs (224 cycles of matrix work
    between request and use), also ACROSS the step boundary: the loader publishes a stage one step early (at the barrier
    that ends step s, stages up to s + 2 have landed), so the first fragments of step s + 1 are requested under the last
    MFMAs of step s and the matrix pipe does not drain at the barrier;
  * every fragment feeds 2 MFMAs (both column tiles), every W fragment MT; an accumulator is touched again 4 MFMAs
    later (64 cycles), beyond the 4-pass MFMA's dependent latency;
  * one s_barrier per step; ~9 scalar instructions per step compute the k step the next W set is loaded from (the
    workgroups' rotated K walk, see linear_wide_kernel).
hipcc's version of this loop (linear_wide_kernel<16, 2, 3, ...>) keeps ONE group of fragments in flight and re-reads after
every barrier: its matrix pipe idles on LDS latency, 28-31 cycles per 16-cycle MFMA (profiles/r05_gemm_wide_streams.json).
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "nano_vllm_amd", "csrc", "gemm_wide_core.inc")

NT = 2          # column tiles per wave
KB = 2          # 32-wide k blocks per step (64 columns)
RING = 4        # W register ring depth (k steps): with the accumulators and the x fragment ring the wave stays within 256
                # registers, so that a fifth wave (the second loader) fits next to a consumer on one SIMD
NS = 4          # LDS stages of the x tile
D = 7           # x fragments requested ahead
R = D + 1       # x fragment ring
L = NT * KB     # W loads per step
ROW_TILE_BYTES = 16 * 64 * 2      # one 16-row tile of a 64-column stage: 2 KiB


def frag_of(f: int):
    """Fragment f of a step -> (row tile, k block). Row tiles go in pairs: (mt0, kb0), (mt1, kb0), (mt0, kb1), (mt1, kb1)."""
    p, u = divmod(f, 4)
    return 2 * p + (u & 1), u >> 1


class Gen:
    def __init__(self, mt: int, skip: str = ""):
        """`skip` (probe builds only): "read" = no x fragment reads, "mfma" = no MFMAs — the rest of the loop unchanged."""
        assert mt % 2 == 0 and (mt * KB) % R == 0 and RING % NS == 0
        self.mt = mt
        self.skip = skip
        self.nfrag = mt * KB
        self.stage_bytes = mt * ROW_TILE_BYTES
        self.lines: list[str] = []

    def emit(self, s: str) -> None:
        if (self.skip == "read" and s.startswith("ds_read")) or (self.skip == "mfma" and s.startswith("v_mfma")):
            return
        self.lines.append(s)

    # operand names
    def acc(self, mt: int, nt: int) -> str:
        return f"%[a{mt * NT + nt}]"

    def w(self, slot: int, nt: int, kb: int) -> str:
        return f"%[w{(slot * NT + nt) * KB + kb}]"

    def x(self, f: int) -> str:
        return f"%[x{f % R}]"

    def read(self, stage: int, f: int) -> str:
        mt, kb = frag_of(f)
        off = stage * self.stage_bytes + mt * ROW_TILE_BYTES
        base = f"%[xa{kb}]" if off < 65536 else f"%[xh{kb}]"
        return f"ds_read_b128 {self.x(f)}, {base} offset:{off % 65536}"

    def prologue(self) -> None:
        e = self.emit
        e("s_nop 4")                                          # "s" operands may come straight from v_readfirstlane
        e("v_add_u32 %[xh0], 0x10000, %[xa0]")
        e("v_add_u32 %[xh1], 0x10000, %[xa1]")
        e("s_sub_u32 %[last], %[steps], 1")
        e("s_mov_b32 %[cnt], 0")
        # W sets of steps 0 .. RING - 2 (clamped to the last step)
        for r in range(RING - 1):
            e(f"s_min_u32 %[t0], {r}, %[last]")
            e("s_add_u32 %[t0], %[t0], %[rot]")
            e("s_cmp_ge_u32 %[t0], %[steps]")
            e("s_cselect_b32 %[t1], %[steps], 0")
            e("s_sub_u32 %[t0], %[t0], %[t1]")
            e("s_mul_i32 %[t0], %[t0], %[wstep]")
            e("v_add_u32 %[vo], %[t0], %[voff]")
            for nt in range(NT):
                for kb in range(KB):
                    e(f"global_load_dwordx4 {self.w(r, nt, kb)}, %[vo], %[wb{nt}] offset:{kb * 1024} nt")
        e("s_barrier")                                        # stages 0 and 1 have landed
        for f in range(D):
            e(self.read(0, f))

    def step(self, i: int, load: bool, vm: int) -> None:
        """Unrolled step i (ring slot i % RING, LDS stage i % NS). `load`: request the W set RING - 1 steps ahead."""
        e = self.emit
        slot, stage = i % RING, i % NS
        e(f"s_waitcnt vmcnt({vm})")
        scalar = []
        if load:
            scalar = [f"s_add_u32 %[t0], %[cnt], {RING - 1}", "s_min_u32 %[t0], %[t0], %[last]", "s_add_u32 %[t0], %[t0], %[rot]",
                      "s_cmp_ge_u32 %[t0], %[steps]", "s_cselect_b32 %[t1], %[steps], 0", "s_sub_u32 %[t0], %[t0], %[t1]",
                      "s_mul_i32 %[t0], %[t0], %[wstep]", "v_add_u32 %[vo], %[t0], %[voff]", "s_add_u32 %[cnt], %[cnt], 1"]
        # W loads of this step: spread over the step, behind the address arithmetic (slots 0 .. 4)
        first = 6
        load_at = {first + j * ((self.nfrag - first) // L): j for j in range(L)} if load else {}
        for f in range(self.nfrag):
            g = f + D
            e(self.read(stage, g) if g < self.nfrag else self.read((stage + 1) % NS, g - self.nfrag))
            e(f"s_waitcnt lgkmcnt({D})")
            mt, kb = frag_of(f)
            for nt in range(NT):
                e(f"v_mfma_f32_16x16x32_bf16 {self.acc(mt, nt)}, {self.w(slot, nt, kb)}, {self.x(f)}, {self.acc(mt, nt)}")
            for _ in range(2):
                if scalar:
                    e(scalar.pop(0))
            if f in load_at:
                j = load_at[f]
                nt, kb = divmod(j, KB)
                e(f"global_load_dwordx4 {self.w((slot + RING - 1) % RING, nt, kb)}, %[vo], %[wb{nt}] offset:{kb * 1024} nt")
        assert not scalar
        e("s_barrier")

    def body(self) -> str:
        e = self.emit
        self.prologue()
        e(f"s_lshr_b32 %[nb], %[steps], {RING.bit_length() - 1}")   # full blocks of RING steps
        e(f"s_and_b32 %[t1], %[steps], {RING - 1}")          # ... and the tail
        e("s_cmp_eq_u32 %[nb], 0")
        e("s_cbranch_scc1 L_tail%=")
        e("L_block%=:")
        for i in range(RING):
            self.step(i, True, (RING - 2) * L)
        e("s_sub_u32 %[nb], %[nb], 1")
        e("s_cmp_lg_u32 %[nb], 0")
        e("s_cbranch_scc1 L_block%=")
        e("L_tail%=:")
        e(f"s_and_b32 %[t1], %[steps], {RING - 1}")
        for j in range(RING - 1):
            e(f"s_cmp_le_u32 %[t1], {j}")
            e("s_cbranch_scc1 L_done%=")
            self.step(j, False, (RING - 2 - j) * L)
        e("L_done%=:")
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e("s_nop 15")                                         # MFMA results -> the compiler's v_accvgpr_read
        e("s_nop 7")
        return "\n".join(f'      "{ln}\\n"' for ln in self.lines)

    def function(self) -> str:
        mt = self.mt
        body = self.body()
        outs = [f'[a{i}] "+a"(acc[{i}])' for i in range(mt * NT)]
        outs += [f'[w{i}] "=&v"(w[{i}])' for i in range(RING * L)]
        outs += [f'[x{i}] "=&v"(x[{i}])' for i in range(R)]
        outs += ['[xh0] "=&v"(xh0)', '[xh1] "=&v"(xh1)', '[vo] "=&v"(vo)', '[t0] "=&s"(t0)', '[t1] "=&s"(t1)', '[cnt] "=&s"(cnt)',
                 '[nb] "=&s"(nb)', '[last] "=&s"(last)']
        ins = ['[xa0] "v"(xa0)', '[xa1] "v"(xa1)', '[voff] "v"(voff)', '[wb0] "s"(wb0)', '[wb1] "s"(wb1)', '[steps] "s"(steps)',
               '[rot] "s"(rot)', '[wstep] "s"(wstep)']

        def wrap(items):
            out, line = [], "      "
            for it in items:
                if len(line) + len(it) > 116:
                    out.append(line.rstrip())
                    line = "      "
                line += it + ", "
            out.append(line.rstrip().rstrip(","))
            return "\n".join(out)

        return f"""// {mt} row tiles x {NT} column tiles per wave, k step 64: {len(self.lines)} instructions
__device__ __forceinline__ void wide_core_mt{mt}{"_no" + self.skip if self.skip else ""}(f32x4_t (&acc)[{mt * NT}], int xa0, int xa1, int voff, uint64_t wb0,
                                                uint64_t wb1, int steps, int rot, int wstep) {{
  u32x4_t w[{RING * L}], x[{R}];
  int xh0, xh1, vo, t0, t1, cnt, nb, last;
  asm volatile(
{body}
      :
{wrap(outs)}
      :
{wrap(ins)}
      : "memory", "scc");
}}
"""


class GenTile:
    """Consumer loop of the FOUR-consumer tile kernel (csrc/gemm_tile4.hip): every SIMD of the CU runs a matrix wave. Wave q
    owns RT row tiles (rows 64 q ...) x ALL CT column tiles of the workgroup; BOTH operands come from LDS — x from the
    LDS-DMA stages of the two x loader waves (4 stages of [256 rows][64 columns], swizzled), W from a 2-stage ring the W
    loader wave fills with ds_write out of its own deep register ring (the HBM latency is hidden there, not in LDS) — so
    no operand is loaded twice and nothing but fragments and accumulators lives in the consumers' registers (<= 256: seven
    waves per CU). One k step = 64 columns = two phases of CT groups of RT MFMAs (phase A: k block 0, phase B: k block 1),
    an s_barrier after each phase:
      barrier a(s), after phase A: x(s + 1) and W(s + 1) have landed  => phase B requests the next step's first fragments;
      barrier b(s), after phase B: every read of x(s) / W(s) has been issued and waited for => their stages are free.
    A W fragment is requested 3 groups before its MFMAs, an x fragment one phase before; lgkmcnt values come from a
    simulation of the in-order LDS queue."""

    def __init__(self, rt: int, ct: int, skip: str = ""):
        assert ct % 2 == 0 and (2 * ct) % 4 == 0 and rt <= ct
        self.rt, self.ct, self.g = rt, ct, 2 * ct
        self.skip = skip                      # probe builds: "read" / "mfma" = the loop without those instructions
        self.lines: list[str] = []
        self.queue: list[str] = []            # LDS reads issued and not known to have completed, oldest first

    def e(self, s: str) -> None:
        if (self.skip == "read" and s.startswith(("ds_read", "s_waitcnt lgkmcnt"))) or (self.skip == "mfma" and s.startswith("v_mfma")):
            return
        self.lines.append(s)

    def acc(self, r: int, c: int) -> str:
        return f"%[a{r * self.ct + c}]"

    def read(self, tag: str, dst: str, addr: str, off: int) -> None:
        self.e(f"ds_read_b128 {dst}, {addr} offset:{off}")
        self.queue.append(tag)

    def wait_for(self, tags) -> None:
        """Everything up to the YOUNGEST of `tags` must have returned: lgkmcnt = reads issued after it."""
        pos = max((i for i, t in enumerate(self.queue) if t in tags), default=-1)
        if pos < 0:
            return
        self.e(f"s_waitcnt lgkmcnt({len(self.queue) - 1 - pos})")
        self.queue = self.queue[pos + 1:]

    def read_w(self, step_tag: str, g: int) -> None:
        kb, c = divmod(g, self.ct)
        self.read(f"{step_tag}w{g}", f"%[w{g % 4}]", "%[wac]", c * 2048 + kb * 1024)

    def phase(self, first: int, tag: str, nxt: str) -> None:
        """Groups first .. first + CT - 1 of step `tag`; `nxt` tags the next step's reads."""
        rt, ct, G = self.rt, self.ct, self.g
        kb = first // ct
        for g in range(first, first + ct):
            c = g - first
            # requests: the W fragment 3 groups ahead (this step's, or the next step's first three once the W address has
            # moved on), and one x fragment of the OTHER k block per group
            if g + 3 < G:
                self.read_w(tag, g + 3)
            else:
                if g + 3 == G:                                # W[G - 1] of this step was requested in the previous group
                    self.e("v_add_u32 %[wac], %[woffn], %[wa]")   # ... from here on: the next step's W stage
                self.read_w(nxt, g + 3 - G)
            if c < rt:
                if kb == 0:
                    self.read(f"{tag}x1_{c}", f"%[fb{c}]", "%[xa1c]", c * 2048)       # k block 1 of this step
                else:
                    self.read(f"{nxt}x0_{c}", f"%[fa{c}]", "%[xa0c]", c * 2048)       # k block 0 of the next step
            self.wait_for({f"{tag}w{g}"} | ({f"{tag}x{kb}_{r}" for r in range(rt)} if c == 0 else set()))
            xr = "fa" if kb == 0 else "fb"
            for r in range(rt):
                self.e(f"v_mfma_f32_16x16x32_bf16 {self.acc(r, c)}, %[w{g % 4}], %[{xr}{r}], {self.acc(r, c)}")
            if kb == 0 and c == 0:                            # stage offsets of the next step (scalar, in the MFMA shadow)
                self.e(f"s_add_u32 %[xoffn], %[xoff], {rt * 4 * 2048}")          # x stage: 4 waves x RT row tiles x 2 KiB
                self.e(f"s_cmp_eq_u32 %[xoffn], {4 * rt * 4 * 2048}")
                self.e("s_cselect_b32 %[xoffn], 0, %[xoffn]")
                self.e(f"s_xor_b32 %[woffn], %[woff], {ct * 2048}")
            if kb == 0 and c == ct - 1:
                self.e("v_add_u32 %[xa0c], %[xoffn], %[xa0]")   # phase B reads the next step's k block 0 fragments
            if kb == 1 and c == ct - 1:
                self.e("v_add_u32 %[xa1c], %[xoffn], %[xa1]")   # the next phase A reads ITS k block 1 fragments
                self.e("s_mov_b32 %[xoff], %[xoffn]")
                self.e("s_mov_b32 %[woff], %[woffn]")
        self.e("s_barrier")

    def body(self) -> str:
        e, rt = self.e, self.rt
        e("s_mov_b32 %[xoff], 0")
        e("s_mov_b32 %[woff], 0")
        e("v_mov_b32 %[xa0c], %[xa0]")
        e("v_mov_b32 %[xa1c], %[xa1]")
        e("v_mov_b32 %[wac], %[wa]")
        e("s_mov_b32 %[cnt], %[steps]")
        e("s_barrier")                                        # x(0), W(0) have landed
        # The first step's leading fragments, requested in the order (and left in the queue state) in which phase B of a
        # step requests the NEXT step's: learnt from a dry run of the loop body.
        dry = GenTile(self.rt, self.ct)
        dry.queue = [f"Sx0_{r}" for r in range(rt)] + [f"Sw{g}" for g in range(3)]
        dry.phase(0, "S", "N")
        dry.phase(self.ct, "S", "N")
        order = [ln for ln in dry.lines if ln.startswith("ds_read")]
        n_next = rt + 3                                       # reads of the next step issued in phase B: its x k-block-0 fragments, W 0..2
        tags_all = []
        probe = GenTile(self.rt, self.ct)
        probe.queue = list(dry.queue)
        # replay phase B's request order for the "N" tags
        seq = []
        d2 = GenTile(self.rt, self.ct)
        d2.queue = [f"Sx0_{r}" for r in range(rt)] + [f"Sw{g}" for g in range(3)]
        d2.phase(0, "S", "N")
        before = len([ln for ln in d2.lines if ln.startswith("ds_read")])
        d2.phase(self.ct, "S", "N")
        # (tags in issue order are not kept by the emitter: rebuild them from the phase-B rules)
        for g in range(self.ct, self.g):
            c = g - self.ct
            if g + 3 >= self.g:
                seq.append(("w", g + 3 - self.g))
            if c < rt:
                seq.append(("x", c))
        assert len(seq) == n_next
        for kind, i in seq:
            if kind == "w":
                self.read_w("S", i)
            else:
                self.read(f"Sx0_{i}", f"%[fa{i}]", "%[xa0c]", i * 2048)
        steady = [t.replace("N", "S", 1) for t in dry.queue]
        assert self.queue[-len(steady):] == steady, (self.queue, steady)
        e(f"s_waitcnt lgkmcnt({len(steady)})")
        self.queue = list(steady)
        entry = list(self.queue)
        e("L_step%=:")
        self.phase(0, "S", "N")
        self.phase(self.ct, "S", "N")
        # the loop-carried LDS queue must look the same on both edges into L_step
        carried = [t.replace("N", "S", 1) for t in self.queue]
        assert carried == entry, (carried, entry)
        e("s_sub_u32 %[cnt], %[cnt], 1")
        e("s_cmp_lg_u32 %[cnt], 0")
        e("s_cbranch_scc1 L_step%=")
        e("s_waitcnt lgkmcnt(0)")
        e("s_nop 15")
        e("s_nop 7")
        return "\n".join(f'      "{ln}\\n"' for ln in self.lines)

    def function(self) -> str:
        rt, ct = self.rt, self.ct
        body = self.body()
        outs = [f'[a{i}] "+a"(acc[{i}])' for i in range(rt * ct)]
        outs += [f'[w{i}] "=&v"(w[{i}])' for i in range(4)]
        outs += [f'[fa{i}] "=&v"(fa[{i}])' for i in range(rt)] + [f'[fb{i}] "=&v"(fb[{i}])' for i in range(rt)]
        outs += [f'[{n}] "=&v"({n})' for n in ("xa0c", "xa1c", "wac")]
        outs += [f'[{n}] "=&s"({n})' for n in ("xoff", "xoffn", "woff", "woffn", "cnt")]
        ins = ['[xa0] "v"(xa0)', '[xa1] "v"(xa1)', '[wa] "v"(wa)', '[steps] "s"(steps)']

        def wrap(items):
            out, line = [], "      "
            for it in items:
                if len(line) + len(it) > 116:
                    out.append(line.rstrip())
                    line = "      "
                line += it + ", "
            out.append(line.rstrip().rstrip(","))
            return "\n".join(out)

        return f"""// {rt} row tiles x {ct} column tiles per wave, both operands from LDS: {len(self.lines)} instructions
__device__ __forceinline__ void tile4_core_r{rt}c{ct}{"_no" + self.skip if self.skip else ""}(f32x4_t (&acc)[{rt * ct}], int xa0, int xa1, int wa, int steps) {{
  u32x4_t w[4], fa[{rt}], fb[{rt}];
  int xa0c, xa1c, wac, xoff, xoffn, woff, woffn, cnt;
  asm volatile(
{body}
      :
{wrap(outs)}
      :
{wrap(ins)}
      : "memory", "scc");
}}
"""


def generate_tile4() -> str:
    head = ("// GENERATED by tools/gen_wide_asm.py — do not edit; class GenTile's docstring describes the schedule.\n\n")
    return head + "\n".join(GenTile(rt, ct).function() for rt, ct in ((4, 6), (4, 8), (4, 4), (3, 6), (3, 8), (3, 4)))


def generate() -> str:
    head = ("// GENERATED by tools/gen_wide_asm.py — do not edit; the generator's docstring describes the schedule.\n"
            f"// NT = {NT}, k blocks per step = {KB}, W ring = {RING} steps, LDS stages = {NS}, x fragments ahead = {D}.\n"
            f"#define NVL_WIDE_CORE_RING {RING}\n#define NVL_WIDE_CORE_STAGES {NS}\n\n")
    return head + "\n".join(Gen(mt).function() for mt in (16, 12))


def generate_probes() -> str:
    """Measurement variants of the 16-row-tile core (probe builds: NVL_PROBES=1 python -m nano_vllm_amd.build): the loop
    without its x fragment reads / without its MFMAs. Results are garbage; tools/gemm_wide_streams.py times them."""
    return ("// GENERATED by tools/gen_wide_asm.py --probes (not committed; probe builds only)\n\n"
            + "\n".join(Gen(16, skip).function() for skip in ("read", "mfma"))
            + "\n".join(GenTile(4, 6, skip).function() for skip in ("read", "mfma")))


if __name__ == "__main__":
    if "--probes" in sys.argv:
        path = OUT.replace("gemm_wide_core.inc", "gemm_wide_core_probes.inc")
        with open(path, "w") as fh:
            fh.write(generate_probes())
        print(path)
        sys.exit(0)
    OUT4 = OUT.replace("gemm_wide_core.inc", "gemm_tile4_core.inc")
    text, text4 = generate(), generate_tile4()
    if "--check" in sys.argv:
        with open(OUT) as fh, open(OUT4) as fh4:
            sys.exit(0 if fh.read() == text and fh4.read() == text4 else 1)
    with open(OUT, "w") as fh:
        fh.write(text)
    with open(OUT4, "w") as fh:
        fh.write(text4)
    print(OUT, len(text), "bytes;", OUT4, len(text4), "bytes")
