"""Build libnvl_hip.so (the C-ABI HIP kernel library) for gfx950 with hipcc.

In-tree build: the .so lands in nano_vllm_amd/lib/ (git-ignored, shipped by gpurun).
hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.

    python -m nano_vllm_amd.build [--force] [--verbose]
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.environ.get("NVL_LIBDIR") or os.path.join(HERE, "lib")   # (NVL_LIBDIR: a probe build next to the shipped one)
LIB = os.path.join(LIBDIR, "libnvl_hip.so")
STAMP = os.path.join(LIBDIR, "libnvl_hip.stamp")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-gpu-rdc"]
if os.environ.get("NVL_PROBES") == "1":       # probe build: the kernels' measurement switches (NVL_WIDE_DBG / _DEBUG) exist
    FLAGS.append("-DNVL_PROBES")
    FLAGS += os.environ.get("NVL_PROBE_FLAGS", "").split()      # e.g. -DNVL_PF_PACKED: a compile-time variant under A/B
    subprocess.run([sys.executable, os.path.join(HERE, "..", "tools", "gen_wide_asm.py"), "--probes"], check=True)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def sources() -> list[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest() -> str:
    h = hashlib.sha256()
    files = sources() + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".inc"))]
    files.append(os.path.join(HERE, "..", "include", "nvl.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def is_fresh() -> bool:
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as fh:
        return fh.read().strip() == _digest()


def _source_digest(src: str) -> str:
    """One translation unit: the source, every header it could include, the flags."""
    h = hashlib.sha256()
    for f in [src] + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".inc"))]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every csrc/*.hip for gfx950 into one shared library; returns its path. Objects are cached per source
    under lib/obj/ (git- and gpurun-ignored), so editing one kernel file recompiles that file only."""
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and is_fresh():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        base = os.path.basename(src)[:-4]
        obj = os.path.join(objdir, base + ".o")
        stamp = os.path.join(objdir, base + ".stamp")
        objs.append(obj)
        digest = _source_digest(src)
        if not force and not verbose and os.path.exists(obj) and os.path.exists(stamp):
            with open(stamp) as fh:
                if fh.read().strip() == digest:
                    continue
        cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, "-c", src, "-o", obj]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd), flush=True)
        procs.append((src, stamp, digest,
                      subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, stamp, digest, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- hipcc failed on {src}\n{out}\n")
            if os.path.exists(stamp):
                os.remove(stamp)
        else:
            with open(stamp, "w") as fh:
                fh.write(digest)
            if verbose or out.strip():
                sys.stderr.write(out)
    if failed:
        raise RuntimeError("hipcc compilation failed")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB]
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as fh:
        fh.write(_digest())
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(path)
