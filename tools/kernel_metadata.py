#!/usr/bin/env python3
"""Per-kernel resource usage of libnerftex_hip.so, read from the code objects embedded in it (no GPU needed).

    python tools/kernel_metadata.py [path/to/lib.so]        -> one line per kernel: vgpr agpr sgpr scratch lds

The library carries one clang offload bundle per translation unit in its `.hip_fatbin` section; each holds the gfx950 ELF whose
`.note` (AMDGPU metadata, msgpack) lists every kernel's `.private_segment_fixed_size` (scratch bytes per lane), register
counts and LDS.  `tests/test_host.py` uses `kernels()` to assert that no MFMA kernel uses scratch."""

from __future__ import annotations

import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib_path: str):
    """(triple, bytes) of every device code object in the library's offload bundles."""
    blob = open(lib_path, "rb").read()
    out = []
    for m in re.finditer(MAGIC, blob):
        base = m.start()
        (n,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        p = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24: p + 24 + idlen].decode()
            p += 24 + idlen
            if size and "amdgcn" in triple:
                out.append((triple, blob[base + off: base + off + size]))
    return out


def kernels(lib_path: str):
    """{demangled-ish kernel name: {vgpr, agpr, sgpr, scratch, lds}} over all code objects of the library."""
    res = {}
    for triple, data in code_objects(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(data); f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        cur = {}
        for line in notes.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s+(.*)$", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip().strip("'")
            if k == "agpr_count" or (k == "args" and cur):
                pass
            if k in ("agpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "vgpr_count", "name", "symbol"):
                cur[k] = v
            if k == "wavefront_size":                       # last key of a kernel's map
                if "name" in cur:
                    res[cur["name"]] = {"vgpr": int(cur.get("vgpr_count", -1)), "agpr": int(cur.get("agpr_count", -1)),
                                        "sgpr": int(cur.get("sgpr_count", -1)), "scratch": int(cur.get("private_segment_fixed_size", -1)),
                                        "lds": int(cur.get("group_segment_fixed_size", -1)), "triple": triple}
                cur = {}
    return res


def pretty(name: str) -> str:
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        return name


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nerf_tex_amd", "libnerftex_hip.so")
    ks = kernels(lib)
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'scratch':>8} {'lds':>7}  kernel")
    for name, k in sorted(ks.items(), key=lambda kv: pretty(kv[0])):
        print(f"{k['vgpr']:>5} {k['agpr']:>5} {k['sgpr']:>5} {k['scratch']:>8} {k['lds']:>7}  {pretty(name)}")
