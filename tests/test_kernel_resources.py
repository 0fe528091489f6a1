"""Static checks on the BUILT gfx950 code objects (no GPU): what the per-kernel AMDGPU metadata inside libnvl_hip.so
says about registers, scratch and LDS. Guards the properties DESIGN.md §3 states and this project's kernels rely on —
a spill, or a kernel that silently lost a wave of occupancy, costs more than most source-level changes gain, and neither
shows up as a test failure anywhere else."""
import os
import sys

import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"),
                                reason="needs the ROCm LLVM binutils (llvm-readelf / llvm-objdump)")


@pytest.fixture(scope="module")
def kernels():
    from nano_vllm_amd import build
    import kernel_resources
    rows = kernel_resources.kernels(build.build())
    assert len(rows) > 100, "could not read the kernel metadata out of libnvl_hip.so"
    return rows


def test_no_kernel_spills_or_uses_scratch(kernels):
    """No kernel touches scratch memory or spills vector registers; no kernel spills scalar registers either, with ONE
    family excepted: the fused decode attention reading its qkv input as fp32 split-K slabs (decode_mfma8_kernel<true, *,
    *, SLABS = true>, round 5; G = 0 is the runtime-group-size instantiation of round 6) sits at the 102-SGPR limit and parks <= 24 scalars in lanes of a vector register
    (v_writelane / v_readlane in the segment prologue, never memory) — the bf16-input instantiations stay clean."""
    def slabs(name):
        return re.search(r"decode_mfma8_kernelILb1ELb[01]ELi[0248]ELb1EE", name) is not None
    bad = [(r["name"], r.get("private_segment_fixed_size"), r.get("vgpr_spill_count"), r.get("sgpr_spill_count"))
           for r in kernels
           if r.get("private_segment_fixed_size", 0) or r.get("vgpr_spill_count", 0)
           or r.get("sgpr_spill_count", 0) > (24 if slabs(r["name"]) else 0)]
    assert not bad, f"kernels with scratch / spills: {bad[:5]}"


def _named(kernels, fragment):
    rows = [r for r in kernels if fragment in r["name"]]
    assert rows, f"no kernel named *{fragment}* in the library"
    return rows


def test_attention_kernels_keep_two_waves_per_simd(kernels):
    """Both attention kernels are launched with 256- or 512-thread workgroups at 2 workgroups (resp. 1) per CU: two
    waves per SIMD need <= 256 unified registers per lane. The decode kernel's fp8 instantiations and the prefill
    kernel's have headroom; none may cross the line."""
    for frag in ("decode_mfma8_kernel", "prefill_attn_kernel", "decode_stream_fp8_kernel"):
        for r in _named(kernels, frag):
            assert r["vgpr_count"] <= 256, (r["name"], r["vgpr_count"])
    # (decode_stream_kernel<8, *>, the packed-dot G = 8 path behind NVL_DECODE_G8_VALU=1, is over the line and runs one
    # wave per SIMD: it is the measured-slower fallback of the matrix-core kernel, not a default path)
    for r in _named(kernels, "decode_stream_kernelILi1E"):
        assert r["vgpr_count"] <= 128, (r["name"], r["vgpr_count"])
    # the prefill kernel's budget after this round's work: <= 216 registers in every instantiation —
    # the zero-fill / waterfall regressions of the past showed up as +20-40 registers first
    for r in _named(kernels, "prefill_attn_kernel"):
        assert r["vgpr_count"] <= 216, (r["name"], r["vgpr_count"])


def test_skinny_gemm_register_budget(kernels):
    """linear_decode_kernel runs 8 waves per workgroup = two per SIMD (<= 256 registers); the instantiations of the
    Qwen3-0.6B decode step (up to 9 row tiles, 4 k-blocks, 2 column tiles) must also leave room for a second workgroup's
    waves on the CU where the grid has more workgroups than CUs (<= 192: at least 2 waves per SIMD with slack)."""
    for r in _named(kernels, "linear_decode_kernel"):
        assert r["vgpr_count"] <= 256, (r["name"], r["vgpr_count"])
    for r in _named(kernels, "linear_decode_kernelILi5ELi4ELi2E"):
        assert r["vgpr_count"] <= 160, (r["name"], r["vgpr_count"])


def test_wide_gemm_uses_the_whole_file_only_with_one_wave_per_simd(kernels):
    """linear_wide_kernel: configurations with 4 consumer waves (+ 2 loaders = 6 waves, max_flat_workgroup_size 384)
    put two waves on some SIMDs and must stay <= 256 registers; the 3-consumer ones (4 waves, 256 threads) own a SIMD
    each and may use the unified 512."""
    for r in _named(kernels, "linear_wide_kernel"):
        limit = 512 if r["max_flat_workgroup_size"] <= 256 else 256
        assert r["vgpr_count"] <= limit, (r["name"], r["vgpr_count"], r["max_flat_workgroup_size"])


def test_norm_and_sampler_kernels_stay_small(kernels):
    """The latency-bound row kernels rely on many resident workgroups: <= 128 registers (the split-K add-RMSNorm keeps up
    to 8 slabs of a row in flight: <= 160)."""
    for frag in ("rmsnorm_kernel", "sample_partial_kernel", "sample_merge_kernel", "decode_stream_combine_kernel",
                 "decode_plan_kernel", "silu_mul_kernel"):
        for r in _named(kernels, frag):
            assert r["vgpr_count"] <= 128, (r["name"], r["vgpr_count"])
    for r in _named(kernels, "add_rmsnorm_splitk_kernel"):
        assert r["vgpr_count"] <= 160, (r["name"], r["vgpr_count"])


def test_prefill_kernel_isa_has_none_of_the_patterns_this_project_removed():
    """Disassembly of the built prefill attention kernels. Each bound below is a pattern hipcc once put into this
    kernel's tile loop and that cost 3-12 % until it was found by reading the .s ("Prefill attention"):
      * a readfirstlane "waterfall" loop around every K/V buffer load (tile coordinates no longer provably uniform):
        s_cbranch_execnz loops and v_readfirstlane counts jump;
      * 49 register moves zero-filling the score tile at every loop head: v_mov_b32 count jumps by ~100 (2 unrolled tiles);
      * an LDS round trip (ds_bpermute) in the per-tile softmax chain: only the two prologue scans may use it;
      * 16 dwordx2 epilogue stores instead of 8 dwordx4; AGPRs used as spill slots (v_accvgpr copies)."""
    from nano_vllm_amd import build
    import kernel_resources
    import re
    dis = kernel_resources.disassemble(build.build(), "prefill_attn_kernel")
    assert len(dis) == 6, sorted(dis)        # {paged, paged fp8, packed} x {4 waves, 8 waves}
    for name, text in dis.items():
        assert text.count("v_mfma_f32_32x32x16_bf16") == 64, name          # two unrolled tiles x 32 MFMAs, nothing else
        assert text.count("s_cbranch_execnz") <= 2, (name, "waterfall loops")
        assert text.count("v_readfirstlane_b32") <= 16, (name, "waterfall loops")
        assert len(re.findall(r"\bv_mov_b32", text)) <= 80, (name, "zero fill of the score registers")
        assert text.count("ds_bpermute_b32") <= 12, (name, "LDS round trip in the tile loop")
        assert text.count("v_accvgpr") == 0 and "scratch_" not in text, (name, "spill copies")
        assert len(re.findall(r"_store_dwordx4", text)) == 8 and not re.findall(r"_store_dwordx2", text), name


def test_generated_prefill_loop_owns_its_register_file(kernels):
    """attn_prefill64.hip runs ONE wave per SIMD on purpose (the whole 512-register file per lane): its generated asm loop
    names v0 .. v191 and a0 .. a223 (tools/gen_prefill_asm.py: scores, fragment ring, O, Q, staged rows). If hipcc's own
    code around the statement ever needed more than the 64 arch registers the clobber list leaves it, it would spill —
    caught by the no-scratch test above; here: the kernel really takes what the stream names and not more than the file."""
    rows = _named(kernels, "prefill_w64_kernel")
    assert len(rows) == 2                                                       # packed K / V, paged cache
    for r in rows:
        assert r.get("agpr_count", 0) >= 224 and r["vgpr_count"] <= 512, (r.get("agpr_count"), r["vgpr_count"])
    core = open(os.path.join(ROOT, "nano_vllm_amd", "csrc", "attn_prefill64_core.inc")).read()
    assert core.count("v_mfma_f32_32x32x16_bf16") == 2 * (32 + 4 * 64 + 2 * 32)  # per stream: first tile, four step bodies, two last steps
    # the steady-state step's budget: <= 7 instructions per MFMA outside the rare rescale path, staging, decisions and
    # barrier included (6.3 without them: the probe's stream)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_prefill_asm as g
    st = g.real_step(0, True, False, "t")
    rare = 2 * (64 * 3 + 32 + 6)            # per block: accvgpr read / mul / write over 64 registers, 32 x shifts, 6 bookkeeping
    hot = len([l for l in st.ins if not l.endswith(":")]) - rare
    assert hot <= 7.0 * 64, hot
