#!/usr/bin/env python3
"""Static audit of the K loops of the GEMM kernels against the assumption their hand-COUNTED `s_waitcnt vmcnt(N)` rest on (advisor,
round 5): a wave's vector-memory queue is in order, so a count is only right while the K loop issues nothing but its LDS-DMA pieces.

The control-flow graph of every k_gemm_pp / k_gemm_dma kernel is rebuilt from the disassembly (branch targets, fall-through) and cut
into natural loops; the K loops are the INNERMOST loops that contain a v_mfma (k_gemm_pp's persistent tile loop and its prologue /
residual prefetch / epilogue lie around them in the graph, wherever the compiler put them in address order).  Inside a K loop every vector-memory instruction must be an LDS-DMA
(`buffer_load_dwordx4 ... lds` or `global_load_lds_dwordx4`): no other load, no store, no atomic, no scratch access -- a
compiler-generated spill or a hoisted / sunk global access would make the counted waits too lax, and stale LDS tiles would be
multiplied without any fault.

    python3 tools/check_loops.py [library]        prints one line per kernel family; exit code 1 on a violation
"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_isa  # noqa: E402

VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "flat_load", "flat_store",
        "flat_atomic", "scratch_load", "scratch_store")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(lib):
    """-> (kernel name, [(byte offset, instruction text, branch-target offset or None)])"""
    for img in check_isa.code_objects(lib):
        path = "/tmp/dh_check_loops.co"
        with open(path, "wb") as f:
            f.write(img)
        txt = subprocess.run([os.path.join(check_isa.LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", path], check=True,
                             capture_output=True, text=True).stdout
        name, base, body = None, 0, []
        for line in txt.splitlines():
            m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
            if m:
                if name:
                    yield name, body
                name, base, body = m.group(2), int(m.group(1), 16), []
                continue
            if not name or not line.startswith("\t"):
                continue
            ins, _, tail = line.partition("//")
            ins = ins.strip()
            am = re.match(r"\s*([0-9A-Fa-f]+):", tail)
            if not am:
                continue
            off = int(am.group(1), 16) - base
            tgt = None
            if ins.startswith(("s_cbranch", "s_branch")):
                tm = re.search(r"<[^>]*\+0x([0-9a-f]+)>", tail)
                tgt = int(tm.group(1), 16) if tm else (0 if re.search(r"<[^+>]+>", tail) else None)
            body.append((off, ins, tgt))
        if name:
            yield name, body


def k_loops(body):
    """The K loops as sets of instruction indices: natural loops of the control-flow graph (a back edge is a branch whose target
    DOMINATES it -- a jump to a block the compiler merely placed at a lower address is not one; body = everything that reaches the
    branch without passing the header; loops of one header are merged) that contain a v_mfma and no other such loop."""
    n = len(body)
    index_of = {off: i for i, (off, _, _) in enumerate(body)}
    ends = ("s_branch", "s_endpgm", "s_setpc_b64")
    leaders = {0}
    for i, (off, ins, tgt) in enumerate(body):
        if tgt is not None and tgt in index_of:
            leaders.add(index_of[tgt])
        if ins.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc_b64")) and i + 1 < n:
            leaders.add(i + 1)
    starts = sorted(leaders)
    block_of = {}
    blocks = []
    for bi, st in enumerate(starts):
        en = (starts[bi + 1] if bi + 1 < len(starts) else n) - 1
        blocks.append((st, en))
        for k in range(st, en + 1):
            block_of[k] = bi
    nb = len(blocks)
    succ = [[] for _ in range(nb)]
    for bi, (st, en) in enumerate(blocks):
        off, ins, tgt = body[en]
        op = ins.split()[0]
        if tgt is not None and tgt in index_of:
            succ[bi].append(block_of[index_of[tgt]])
        if op not in ends and en + 1 < n:
            succ[bi].append(block_of[en + 1])
    pred = [[] for _ in range(nb)]
    for u, ss in enumerate(succ):
        for v in ss:
            pred[v].append(u)
    # reverse postorder from the entry, then Cooper / Harvey / Kennedy immediate dominators
    order, seen, stack = [], {0}, [(0, iter(succ[0]))]
    while stack:
        u, it = stack[-1]
        for v in it:
            if v not in seen:
                seen.add(v)
                stack.append((v, iter(succ[v])))
                break
        else:
            order.append(u)
            stack.pop()
    rpo = order[::-1]
    num = {b: i for i, b in enumerate(rpo)}
    idom = {0: 0}
    changed = True
    while changed:
        changed = False
        for b in rpo[1:]:
            new = None
            for q in pred[b]:
                if q not in idom:
                    continue
                if new is None:
                    new = q
                else:
                    x, y = q, new
                    while x != y:
                        while num[x] > num[y]:
                            x = idom[x]
                        while num[y] > num[x]:
                            y = idom[y]
                    new = x
            if new is not None and idom.get(b) != new:
                idom[b] = new
                changed = True

    def dominates(a, b):
        while True:
            if a == b:
                return True
            if b == 0 or b not in idom:
                return False
            b = idom[b]

    loops = {}
    for u in rpo:
        for h in succ[u]:
            if h in num and dominates(h, u):
                members, stack2 = {h, u}, [u]
                while stack2:
                    k = stack2.pop()
                    if k == h:
                        continue
                    for q in pred[k]:
                        if q in num and q not in members:
                            members.add(q)
                            stack2.append(q)
                loops.setdefault(h, set()).update(members)

    def has_mfma(bs):
        return any(body[k][1].startswith("v_mfma") for bi in bs for k in range(blocks[bi][0], blocks[bi][1] + 1))

    with_mf = {h: m for h, m in loops.items() if has_mfma(m)}
    inner = [m for h, m in sorted(with_mf.items()) if not any(o != h and o in m for o in with_mf)]
    return [{k for bi in m for k in range(blocks[bi][0], blocks[bi][1] + 1)} for m in inner]


def audit(name, body):
    """-> (number of K loops, MFMAs inside them, LDS-DMA instructions inside them, [violations])"""
    loops = k_loops(body)
    bad, n_mf, n_dma = [], 0, 0
    for members in loops:
        for k in sorted(members):
            off, ins, _ = body[k]
            op = ins.split()[0]
            if op.startswith("v_mfma"):
                n_mf += 1
            if op.startswith(VMEM):
                dma = op == "global_load_lds_dwordx4" or (op == "buffer_load_dwordx4" and ins.rstrip().endswith(" lds"))
                if dma:
                    n_dma += 1
                else:
                    bad.append((hex(off), ins))
    return len(loops), n_mf, n_dma, bad


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "diffusionhandles_amd", "libdiffhandles_hip.so")
    fams, rc = {}, 0
    for name, body in kernels(lib):
        if "k_gemm_pp" not in name and "k_gemm_dma" not in name:
            continue
        nl, n_mf, n_dma, bad = audit(name, body)
        fam = "k_gemm_pp" if "k_gemm_pp" in name else "k_gemm_dma"
        f = fams.setdefault(fam, [0, 0, 0, 0, 0])
        f[0] += 1; f[1] += nl; f[2] += n_mf; f[3] += n_dma; f[4] += len(bad)
        if nl == 0 or n_dma == 0:
            print("NO K LOOP FOUND (or one without LDS-DMA):", name, nl, n_mf, n_dma)
            rc = 1
        for b in bad:
            print("VIOLATION", name, *b)
            rc = 1
    for fam, f in sorted(fams.items()):
        print(f"{fam}: {f[0]} kernels, {f[1]} K loops, {f[2]} MFMAs and {f[3]} LDS-DMA instructions inside them, {f[4]} other vector-memory instructions")
    return rc


if __name__ == "__main__":
    sys.exit(main())
