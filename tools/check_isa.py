#!/usr/bin/env python3
"""Register / scratch audit of every kernel in libdiffhandles_hip.so, from the code-object metadata (no GPU needed).

  python tools/check_isa.py [lib.so]      # table: kernel, VGPRs, AGPRs, SGPRs, scratch bytes, LDS bytes, max threads

Used by tests/test_abi.py to keep out the regressions an ISA pass found in round 1: a kernel that silently needs
scratch (address-taken locals, spills) or so many registers that only one wave fits a SIMD where several should.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib):
    """The gfx950 ELF images of every offload bundle in the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(td, "x")],
                       check=True, capture_output=True)
        data = open(fat, "rb").read()
    out, pos = [], 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            break
        (n,) = struct.unpack_from("<Q", data, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", data, q)
            ident = data[q + 24:q + 24 + idlen].decode()
            q += 24 + idlen
            if "gfx950" in ident and size:
                out.append(data[pos + off:pos + off + size])
        pos += len(MAGIC)
    return out


def kernels(lib):
    """[{name, vgpr, agpr, sgpr, scratch, lds, max_threads}]"""
    res = []
    for img in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], check=True, capture_output=True,
                                 text=True).stdout
        for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
            blk = ".agpr_count:" + blk
            g = lambda key: re.search(r"\.%s:\s*(\S+)" % key, blk)
            res.append(dict(name=g("name").group(1), vgpr=int(g("vgpr_count").group(1)), agpr=int(g("agpr_count").group(1)),
                            sgpr=int(g("sgpr_count").group(1)), scratch=int(g("private_segment_fixed_size").group(1)),
                            lds=int(g("group_segment_fixed_size").group(1)), max_threads=int(g("max_flat_workgroup_size").group(1))))
    return res


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                              "diffusionhandles_amd", "libdiffhandles_hip.so")
    ks = kernels(lib)
    print(f"{len(ks)} kernels")
    for k in sorted(ks, key=lambda k: -(k["vgpr"] + k["agpr"])):
        print(f"{k['vgpr']:4d} v {k['agpr']:4d} a {k['sgpr']:4d} s  scratch {k['scratch']:4d}  lds {k['lds']:7d}  thr {k['max_threads']:5d}  {k['name'][:100]}")
