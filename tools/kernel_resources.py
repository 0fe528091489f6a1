"""Per-kernel register / scratch / LDS budget of the BUILT library, read from the code objects inside libnvl_hip.so
(no recompilation): the `.hip_fatbin` section holds one clang offload bundle per translation unit; each gfx950 entry is
an ELF whose AMDGPU metadata note lists every kernel's vgpr / agpr / sgpr count, spill counts, private (scratch) and
group (static LDS) segment sizes. Usage: python tools/kernel_resources.py [lib.so] [--json]"""
from __future__ import annotations

import json
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _fatbin(path: str) -> bytes:
    out = subprocess.run([f"{LLVM}/llvm-readelf", "-S", "-W", path], capture_output=True, text=True, check=True).stdout
    for line in out.splitlines():
        m = re.search(r"\.hip_fatbin\s+\w+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", line)
        if m:
            off, size = int(m.group(2), 16), int(m.group(3), 16)
            with open(path, "rb") as fh:
                fh.seek(off)
                return fh.read(size)
    raise RuntimeError(f"{path}: no .hip_fatbin section")


def code_objects(path: str, arch: str = "gfx950") -> list[bytes]:
    blob, objs, pos = _fatbin(path), [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if arch in triple and size:
                objs.append(blob[pos + off:pos + off + size])
        pos += len(MAGIC)
    return objs


def kernels(path: str) -> list[dict]:
    rows = []
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as tf:
            tf.write(co)
            tf.flush()
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", tf.name], capture_output=True, text=True).stdout
        cur: dict = {}
        for line in notes.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
            if not m:
                continue
            key, val = m.group(1), m.group(2).strip().strip("'")
            if key == "agpr_count" and cur.get("name"):          # first key of a kernel entry (alphabetical order)
                rows.append(cur)
                cur = {}
            if key in ("name", "symbol"):
                cur[key] = val
            elif key in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                         "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size"):
                cur[key] = int(val)
        if cur.get("name"):
            rows.append(cur)
    filt = next((c for c in (f"{LLVM}/llvm-cxxfilt", "/usr/bin/c++filt") if os.path.exists(c)), None)
    names = [r["name"] for r in rows]
    if filt:
        names = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for r, d in zip(rows, names):
        r["demangled"] = d.replace("(anonymous namespace)::", "")
    return rows


def disassemble(path: str, fragment: str) -> dict[str, str]:
    """{kernel symbol: disassembly text} of every kernel of the library whose (mangled) name contains `fragment`."""
    out = {}
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as tf:
            tf.write(co)
            tf.flush()
            syms = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "-W", tf.name], capture_output=True, text=True).stdout
            names = sorted({ln.split()[-1] for ln in syms.splitlines()
                            if " FUNC " in ln and fragment in ln and not ln.split()[-1].endswith(".kd")})
            if not names:
                continue
            text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--mcpu=gfx950",
                                   "--disassemble-symbols=" + ",".join(names), tf.name],
                                  capture_output=True, text=True).stdout
        cur = None
        for ln in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
            if m:
                cur = m.group(1) if m.group(1) in names else None
                if cur:
                    out[cur] = ""
            elif cur:
                out[cur] += ln + "\n"
    return out


if __name__ == "__main__":
    lib = next((a for a in sys.argv[1:] if not a.startswith("--")),
               os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nano_vllm_amd", "lib",
                            "libnvl_hip.so"))
    rows = kernels(lib)
    if "--json" in sys.argv:
        print(json.dumps(rows))
    else:
        for r in sorted(rows, key=lambda r: r["demangled"]):
            print(f'{r.get("vgpr_count", 0):4d} v {r.get("agpr_count", 0):4d} a {r.get("private_segment_fixed_size", 0):5d} B scratch '
                  f'{r.get("vgpr_spill_count", 0):3d} spills {r.get("group_segment_fixed_size", 0):6d} B lds  {r["demangled"][:110]}')
        print(f"{len(rows)} kernels; spilling: {sum(1 for r in rows if r.get('vgpr_spill_count', 0) or r.get('private_segment_fixed_size', 0))}")
