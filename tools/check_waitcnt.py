"""Static check of the hand-scheduled LDS reads in the gfx950 code objects (CPU only: no GPU needed).

The transform and fused-layer kernels issue ``ds_read_b128`` / ``ds_read_b32`` from inline asm and cover them
with counted ``s_waitcnt lgkmcnt(N)`` placed by hand; the compiler sees neither and may schedule a use of a
fragment register - or a copy of it - above the wait (commit 7606cc0: an MFMA read a fragment that had not
landed, one run in five).  This walks the DISASSEMBLY of every kernel (``llvm-objdump -d`` of the device code
object embedded in a built ``.o``), follows the control flow, tracks the LDS / scalar-memory operations that are
still outstanding at every instruction, and reports every instruction that touches a vector register whose
``ds_read`` is not yet covered by a wait.

Model (CDNA ISA, ``LGKM_CNT``): LDS operations of a wave return in issue order, so ``lgkmcnt(N)`` guarantees all
but the N youngest; scalar-memory loads share the counter and return out of order, so while one is outstanding
only ``lgkmcnt(0)`` guarantees anything.  Where two paths with different outstanding sets meet, every register
pending on either stays pending, with the smaller number of younger LDS operations behind it (conservative).

    python tools/check_waitcnt.py primekg_rgcn_linkprediction_amd/csrc/build/*.o
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile
from typing import Dict, FrozenSet, List, Optional, Tuple  # noqa: F401

LLVM_BIN = "/opt/rocm/lib/llvm/bin"

_FUNC = re.compile(r"^([0-9a-f]+) <(.+)>:\s*$")
_INST = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_TARGET = re.compile(r"<(.+)\+0x([0-9a-f]+)>\s*$")
_REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")
_LGKM = re.compile(r"lgkmcnt\((\d+)\)")
_VM = re.compile(r"vmcnt\((\d+)\)")


class Inst:
    __slots__ = ("addr", "op", "args", "text", "target")

    def __init__(self, addr: int, op: str, args: str, text: str, target: Optional[int]):
        self.addr, self.op, self.args, self.text, self.target = addr, op, args, text, target


def _regs(text: str) -> FrozenSet[str]:
    out = set()
    for kind, single, lo, hi in _REG.findall(text):
        if single:
            out.add(kind + single)
        else:
            out.update(kind + str(i) for i in range(int(lo), int(hi) + 1))
    return frozenset(out)


def parse_disassembly(text: str) -> Dict[str, List[Inst]]:
    """``llvm-objdump -d`` output -> {function: [Inst, ...]} (instructions in address order)"""
    funcs: Dict[str, List[Inst]] = {}
    base: Dict[str, int] = {}
    cur = None
    for line in text.splitlines():
        m = _FUNC.match(line)
        if m:
            cur = m.group(2)
            funcs[cur] = []
            base[cur] = int(m.group(1), 16)
            continue
        if cur is None:
            continue
        m = _INST.match(line)
        if not m:
            continue
        op, args, addr = m.group(1), m.group(2), int(m.group(3), 16)
        target = None
        if op.startswith("s_cbranch") or op == "s_branch":
            t = _TARGET.search(line)
            if t and t.group(1) in base:
                target = base[t.group(1)] + int(t.group(2), 16)
            else:                                    # no symbolic target printed: simm16 words from the next instruction
                off = int(args.split()[0])
                off = off - 65536 if off >= 32768 else off
                target = addr + 4 + 4 * off
        funcs[cur].append(Inst(addr, op, args, line.strip(), target))
    return funcs


def _is_lds(op: str) -> bool:
    return op.startswith("ds_")


def _is_smem(op: str) -> bool:
    return op.startswith(("s_load", "s_buffer_load", "s_store", "s_buffer_store", "s_memtime", "s_memrealtime",
                          "s_atomic", "s_buffer_atomic", "s_dcache", "s_sendmsg"))


def _is_vmem(op: str) -> bool:
    return op.startswith(("global_", "buffer_", "flat_", "scratch_", "tbuffer_"))


def _vmem_dest(op: str, args: str) -> FrozenSet[str]:
    """vector registers a vector-memory operation writes when its data returns (LDS-DMA loads write none)"""
    if "load" not in op or "_lds_" in op:
        return frozenset()
    return _regs(args.split(",")[0])


def _lds_dest(op: str, args: str) -> FrozenSet[str]:
    """vector registers an LDS operation writes when its data returns"""
    returning = op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append",
                               "ds_ordered_count")) or "_rtn" in op
    if not returning:
        return frozenset()
    first = args.split(",")[0]
    return _regs(first)


# State at a program point: {register: age} for every vector register an LDS read still has to deliver into, where
# age = the number of LDS operations issued AFTER that read on every path to this point (the minimum over the paths:
# smaller is harder to retire, so the minimum is the conservative merge), plus one flag: a scalar-memory operation
# may be outstanding (then only lgkmcnt(0) proves anything).  lgkmcnt(N), N > 0, retires the reads with age >= N:
# LDS data returns in issue order and at most N operations are still outstanding, so a read with N younger ones
# behind it is not among them.
_AGE_CAP = 64


def _merge(a, b):
    if a is None:
        return b
    regs = dict(a[0])
    for r, age in b[0].items():
        regs[r] = min(age, regs[r]) if r in regs else age
    return regs, a[1] or b[1]


def check_function(insts: List[Inst]) -> List[str]:
    """-> one line per instruction that touches a register with an uncovered ds_read - or an uncovered vector-memory
    load (``VM_CNT``: round 4's transform kernels request operand maxima with ``global_load_dword`` from inline asm
    ahead of their LDS-DMAs and cover them with a counted ``s_waitcnt vmcnt(N)``) - (empty: clean)"""
    return _check(insts, "lgkm") + _check(insts, "vm")


def _check(insts: List[Inst], counter: str) -> List[str]:
    """one counter's pass: "lgkm" (LDS reads; scalar memory makes the counter out of order) or "vm" (vector-memory
    loads, in issue order with every other vector-memory operation, LDS-DMAs and stores included - the model the
    compiler itself uses on gfx9)"""
    if not insts:
        return []
    vm = counter == "vm"
    wait_re = _VM if vm else _LGKM
    is_op = _is_vmem if vm else _is_lds
    dest_of = _vmem_dest if vm else _lds_dest
    index = {inst.addr: i for i, inst in enumerate(insts)}
    leaders = {0}
    for i, inst in enumerate(insts):
        if inst.target is not None:
            if inst.target in index:
                leaders.add(index[inst.target])
            if i + 1 < len(insts):
                leaders.add(i + 1)
        elif inst.op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64") and i + 1 < len(insts):
            leaders.add(i + 1)
    order = sorted(leaders)
    block_end = {b: (order[k + 1] if k + 1 < len(order) else len(insts)) for k, b in enumerate(order)}
    entry = {0: ({}, False)}
    work = [0]
    bad: Dict[int, str] = {}
    while work:
        b = work.pop()
        regs, smem = dict(entry[b][0]), entry[b][1]
        succ: List[int] = []
        fall = True
        for i in range(b, block_end[b]):
            inst = insts[i]
            op = inst.op
            if op == "s_waitcnt":
                m = wait_re.search(inst.args)
                if m:
                    n = int(m.group(1))
                    if n == 0:
                        regs, smem = {}, False
                    elif not smem:
                        regs = {r: age for r, age in regs.items() if age < n}
                continue
            if is_op(op):
                dest = dest_of(op, inst.args)
                srcs = _regs(inst.args.split(",", 1)[1]) if (dest and "," in inst.args) else (_regs(inst.args) - dest)
                hit = srcs & regs.keys()
                if hit:
                    bad[inst.addr] = f"{inst.text}   <- {sorted(hit)} not covered by a wait"
                regs = {r: min(age + 1, _AGE_CAP) for r, age in regs.items()}
                for r in dest:                       # in-order returns: a queued register is simply delivered again
                    regs[r] = 0
                continue
            if _is_smem(op) and not vm:
                smem = True
                continue
            hit = _regs(inst.args) & regs.keys()
            if hit:
                bad[inst.addr] = f"{inst.text}   <- {sorted(hit)} not covered by a wait"
            if inst.target is not None:
                if inst.target in index:
                    succ.append(index[inst.target])
                if op == "s_branch":
                    fall = False
                break
            if op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
                fall = False
                break
        if fall and block_end[b] < len(insts):
            succ.append(block_end[b])
        out = (regs, smem)
        for t in succ:
            merged = _merge(entry.get(t), out)
            if t not in entry or merged != entry[t]:
                entry[t] = merged
                work.append(t)
    return [bad[a] for a in sorted(bad)]


def disassemble_object(path: str) -> str:
    """the gfx950 code object inside a hipcc-built ``.o`` (or a bare code object) -> ``llvm-objdump -d`` text"""
    objdump = os.path.join(LLVM_BIN, "llvm-objdump")
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(path))
        with open(path, "rb") as src, open(local, "wb") as dst:
            dst.write(src.read())
        subprocess.run([objdump, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        device = [f for f in os.listdir(tmp) if "amdgcn" in f]
        target = os.path.join(tmp, device[0]) if device else local
        return subprocess.run([objdump, "-d", target], check=True, capture_output=True, text=True).stdout


def check_object(path: str) -> Dict[str, List[str]]:
    """-> {kernel: [violations]} for the kernels that have any; also counts via the second return of ``stats``"""
    funcs = parse_disassembly(disassemble_object(path))
    return {name: v for name, insts in funcs.items() for v in [check_function(insts)] if v}


def stats(path: str) -> Tuple[int, int]:
    """(kernels, ds_read instructions) seen in the object - so that a test can tell "clean" from "nothing parsed" """
    funcs = parse_disassembly(disassemble_object(path))
    return len(funcs), sum(1 for insts in funcs.values() for i in insts if i.op.startswith("ds_read"))


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        found = check_object(p)
        k, reads = stats(p)
        print(f"{p}: {k} kernels, {reads} ds_read instructions, {sum(len(v) for v in found.values())} violations")
        for name, lines in found.items():
            rc = 1
            print(" ", name)
            for line in lines:
                print("    ", line)
    sys.exit(rc)
