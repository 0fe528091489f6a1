"""CPU oracle for the nano-vllm hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package, and only as the checker. Nothing under `nano_vllm_amd/` or `nanovllm/` imports it.

What it is: a plain PyTorch-CPU (fp32 math) restatement of the arithmetic the reference
(GeeeekExplorer/nano-vllm v0.2.0) performs on the paged-KV forward step. Every function cites
the reference file:line it follows. The reference delegates most of this arithmetic to
third-party packages that are NOT under /root/reference:

  * flash-attn (unpinned, pyproject.toml:18)      -> oracle.ops.flash_attn_varlen_func /
                                                      flash_attn_with_kvcache restate its
                                                      published semantics (see their docstrings)
  * triton>=3.0.0 (pyproject.toml:16)             -> kernel source is in tree (attention.py:10-30)
  * torch.compile / inductor (torch>=2.4.0)       -> rounding points of the compiled graphs
  * xxhash (unpinned, pyproject.toml:19)          -> imported directly (it is installed)

One file restates the PRODUCT, not the reference: `oracle/philox.py` is an independent numpy restatement of the sampler's
counter-based draw (Philox4x32-10 keyed by seed / request ordinal / position / column), pinned to the published
Random123 known-answer vectors and to the library's own host replay. The reference's draw (torch's generator consumed
in batch order, sampler.py:11) is not reproducible even between the reference's eager and compiled forms; with the
product's draw restated, `oracle/judge.py` can judge sampled tokens EXACTLY on the oracle's logits
(argmax of l/T - log E) instead of only distributionally.

Pinning status: the reference ships no tests and no golden vectors, so parity
is pinned by outputs of the reference ITSELF, produced in the build container by importing
/root/reference with a `flash_attn` stub (oracle/ref_import.py, oracle/make_golden.py) and
committed under tests/golden/. At the flash-attn boundary the stub IS this restatement, so
there parity is "unpinned" (flash-attn cannot be installed or run here) — stated in DESIGN.md.
"""
