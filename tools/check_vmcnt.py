#!/usr/bin/env python3
"""Static check of hand-placed vector-memory waits (no GPU needed): a linear scoreboard over the disassembly of every kernel of
libdiffhandles_hip.so.

Some kernels issue global loads from inline asm into "=&v" registers and wait for them with a hand-written s_waitcnt (the
attention kernels' tile prefetch, csrc/attention.hip fetch_tile / tile_wait): the compiler's own wait insertion does not know
those loads, so a copy or spill of their destination registers between the load and the wait would read a register with a load
still in flight -- silently.  This walks each kernel's instructions in address order, keeps the vector-memory operations in flight
in issue order (loads AND stores: they share the in-order vmcnt counter on gfx9), retires them at every `s_waitcnt vmcnt(N)` and
reports any instruction that touches a destination register of a load still in flight.  Straight-line approximation: conditional
branches fall through, the scoreboard is cleared at unconditional branches and at s_endpgm, loop back-edges are not followed.

    python tools/check_vmcnt.py [lib.so] [kernel-name-fragment ...]       exit code 1 when a violation is found"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_isa

VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "flat_load", "flat_store",
        "flat_atomic", "scratch_load", "scratch_store")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def kernels_disassembly(lib):
    for img in check_isa.code_objects(lib):
        path = "/tmp/dh_check_vmcnt.co"
        with open(path, "wb") as f:
            f.write(img)
        txt = subprocess.run([os.path.join(check_isa.LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", path], check=True,
                             capture_output=True, text=True).stdout
        name, body = None, []
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                if name:
                    yield name, body
                name, body = m.group(1), []
            elif name and line.startswith("\t"):
                body.append(line.split("//")[0].strip())
        if name:
            yield name, body


def check(name, body):
    """-> list of (instruction index, instruction, in-flight load) violations"""
    flight = []                     # (instruction text, set of destination registers) in issue order
    bad = []
    for idx, ins in enumerate(body):
        mn = ins.split()[0] if ins else ""
        if mn == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if m:
                n = int(m.group(1))
                flight = flight[len(flight) - n:] if n and n < len(flight) else ([] if n == 0 else flight)
            continue
        if mn in ("s_endpgm", "s_branch", "s_setpc_b64"):
            flight = []
            continue
        touched = regs_of(ins)
        if mn.startswith(VMEM):
            ops = ins[len(mn):].split(",")
            is_load = "_load" in mn or ("atomic" in mn and " glc" in ins)
            to_lds = "_lds_" in mn or re.search(r"\blds\b", ins) is not None
            dst = regs_of(ops[0]) if is_load and not to_lds else set()
            src = regs_of(",".join(ops[1:])) if dst else touched
            for f_ins, f_dst in flight:
                if f_dst & src:                     # (a second load into the same register is not a hazard: returns are in order)
                    bad.append((idx, ins, f_ins))
            flight.append((ins, dst))
            if len(flight) > 63:
                flight = flight[-63:]               # the counter saturates: older operations have retired
            continue
        for f_ins, f_dst in flight:
            if f_dst & touched:
                bad.append((idx, ins, f_ins))
                break
    return bad


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffusionhandles_amd", "libdiffhandles_hip.so")
    frags = [a for a in sys.argv[1:] if not a.endswith(".so")]
    nk = nbad = 0
    for name, body in kernels_disassembly(lib):
        if frags and not any(f in name for f in frags):
            continue
        nk += 1
        bad = check(name, body)
        if bad:
            nbad += 1
            print(f"{name}: {len(bad)} instruction(s) touch a register with a load in flight")
            for idx, ins, f_ins in bad[:5]:
                print(f"    #{idx}: {ins}    <- in flight: {f_ins}")
    print(f"{nk} kernels checked, {nbad} with violations")
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main())
