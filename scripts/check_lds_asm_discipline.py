#!/usr/bin/env python3
"""Binary lint for the hand-scheduled LDS reads of the HIP kernels (build-time check, CPU only).

The hub kernels of the aggregation (gnnx_spmm.hip) issue their LDS reads from inline asm and cover them with hand-counted
`s_waitcnt lgkmcnt(N)` statements, so that a set of reads stays in flight while the previous set is added.  The compiler knows
nothing of that: to it the output registers of a `ds_read` asm are defined when the statement ends, and it may COPY or overwrite
them (a loop-carried value, a register shuffle) before the wait that covers the load -- the copy then takes whatever the register
held before the data arrived.  That is not hypothetical: round 5's first edit of spmm_hubpc_kernel (an early return inside the
consumer's loop) made the register allocator copy a whole in-flight set at the loop's back edge, and the kernel returned
run-to-run different sums on every hub row; the unedited source had the same latent property and was only correct as compiled.

This script makes the property a CHECKED one: it disassembles the gfx950 code object of an object file and walks every path of
the chosen kernels with the queue of outstanding LGKM operations (LDS reads and writes, scalar loads) as state:
  * `ds_read*` pushes its destination registers, every other LDS operation an entry without registers;
  * `s_waitcnt ... lgkmcnt(N)` keeps the N youngest entries: LDS operations return in order, and scalar loads -- which share the
    counter and return out of order -- can only make the wait stronger for the LDS operations (pending LDS <= pending total <= N),
    so they are left out of the queue;
  * any other instruction that names a register of an entry still in the queue -- as source or destination -- is a violation.
Exit status 1 and one line per violation (kernel, address, instruction, the read it collides with).

usage: check_lds_asm_discipline.py OBJECT.o [--kernels REGEX] [--max-states N]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)")   # v12, v[4:7], a3 (AGPRs share the hazard)


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        kind = m.group(1)
        if m.group(4) is not None:
            out.add((kind, int(m.group(4))))
        else:
            out.update((kind, r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def disassemble(obj):
    """-> {function name: [(addr, mnemonic, operands text)]} of the gfx950 code object bundled in `obj`."""
    with tempfile.TemporaryDirectory() as tmp:
        base = os.path.join(tmp, os.path.basename(obj))
        with open(obj, "rb") as f, open(base, "wb") as g:
            g.write(f.read())
        subprocess.run([OBJDUMP, "--offloading", base], cwd=tmp, check=True, capture_output=True)
        cos = [os.path.join(tmp, n) for n in os.listdir(tmp) if "amdgcn" in n]
        if not cos:
            raise SystemExit(f"no gfx code object found in {obj}")
        text = subprocess.run([OBJDUMP, "-d", cos[0]], check=True, capture_output=True, text=True).stdout
    funcs, cur = {}, None
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", ln)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return funcs


def successors(ins, i):
    """indices of the instructions that may follow ins[i]"""
    addr, mn, ops = ins[i]
    by_addr = successors.index
    if mn == "s_endpgm":
        return []
    if mn == "s_branch" or mn.startswith("s_cbranch"):
        imm = int(ops.split()[0])
        if imm >= 32768:
            imm -= 65536
        tgt = by_addr.get(addr + 4 + 4 * imm)
        if tgt is None:
            raise SystemExit(f"branch at {addr:x} leaves the function")
        return [tgt] if mn == "s_branch" else [tgt, i + 1]
    return [i + 1] if i + 1 < len(ins) else []


def check(name, ins, max_states):
    successors.index = {a: k for k, (a, _, _) in enumerate(ins)}
    bad = {}
    # state: (index, queue) with queue = tuple of entries (issue address, frozenset of registers, is_scalar_load)
    start = (0, ())
    seen, work = {start}, [start]
    while work:
        if len(seen) > max_states:
            raise SystemExit(f"{name}: more than {max_states} states -- raise --max-states")
        i, q = work.pop()
        addr, mn, ops = ins[i]
        if mn == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", ops)
            if m:
                n = int(m.group(1))
                q = q[len(q) - n:] if n < len(q) else q
        else:
            touched = regs_of(ops)
            for e in q:
                hit = touched & e[1]
                if hit:
                    k, r = sorted(hit)[0]
                    bad.setdefault((addr, e[0]), f"{name}: {addr:x}  {mn} {ops}   touches {k}{r}, the destination of the LDS read at {e[0]:x} "
                                                 f"that no s_waitcnt lgkmcnt has covered yet")
            if mn.startswith("ds_read"):
                dst = ops.split(",")[0]
                q = q + ((addr, frozenset(regs_of(dst)), False),)
            elif mn.startswith("ds_write") or mn.startswith("ds_"):   # other LDS operations count in lgkmcnt too
                q = q + ((addr, frozenset(), False),)
            if len(q) > 64:
                q = q[-64:]
        for j in successors(ins, i):
            st = (j, q)
            if st not in seen:
                seen.add(st)
                work.append(st)
    return list(bad.values()), len(seen)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("obj")
    ap.add_argument("--kernels", default=r"spmm_hubpc_kernel|spmm_hub_kernel")
    ap.add_argument("--max-states", type=int, default=2_000_000)
    args = ap.parse_args()
    funcs = disassemble(args.obj)
    pat = re.compile(args.kernels)
    picked = {n: ins for n, ins in funcs.items() if pat.search(n) and ins}
    if not picked:
        raise SystemExit(f"no kernel of {args.obj} matches {args.kernels}")
    total = []
    for n, ins in sorted(picked.items()):
        v, states = check(n, ins, args.max_states)
        total += v
        print(f"{'FAIL' if v else 'ok  '} {n}: {len(ins)} instructions, {states} states, {len(v)} violations")
    for line in total[:40]:
        print(line)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
