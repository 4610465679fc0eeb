#!/usr/bin/env python3
"""Build-time check of the emitted gfx950 ISA of k_partials_mfma_wide (csrc/hip/kernels_mfma_wide.h).

That kernel issues its child loads through inline asm and waits for them with hand-counted `s_waitcnt vmcnt(N)`
(the compiler's own bookkeeping waits for everything in flight at every loop back edge). It is correct only if
NOTHING touches a load's destination registers between the load and the wait that retires it - and that is a
property of the compiled code, not of the source: a register copy at the loop's back edge, a v_mov, a spill to
accumulation registers (v_accvgpr_write on the unified file) or to scratch would read a register before its load
has landed and give silently wrong, run-to-run different CLVs. The host-side check (`wide_kernel_is_sound`,
pllgpu.hip) sees scratch only. This script replays every instantiation's instruction stream:

  * a FIFO of the wave's outstanding vector-memory operations, in issue order (loads, stores and atomics count
    together on gfx9, MI355X_MICROARCH.md "s_waitcnt"); `s_waitcnt vmcnt(N)` retires all but the youngest N;
  * any instruction that names a register (v or a) that is still the destination of an outstanding load fails the
    build, as does any scratch_* / flat_* instruction (memory operations the hand count does not know, or that
    return out of order);
  * a backward branch replays its loop body once more with the state at the branch (the counted waits must hold
    across the back edge: that is where the first item's loads meet the second item's waits).

usage: check_wide_isa.py <pllgpu.o> [--kernel SUBSTRING] [--verbose]      exit code 0 = sound
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
VMEM_LOAD = re.compile(r"^(global_load|buffer_load|scratch_load|flat_load)")
VMEM_OTHER = re.compile(r"^(global_store|buffer_store|global_atomic|buffer_atomic|buffer_wbl2|buffer_inv)")
FORBIDDEN = re.compile(r"^(scratch_|flat_)")
REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(2) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(1), i) for i in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


def disassemble(obj):
    tmp = tempfile.mkdtemp(prefix="wide_isa_")
    try:
        local = os.path.join(tmp, "unit.o")
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True)
        cos = [f for f in os.listdir(tmp) if "gfx950" in f]
        if len(cos) != 1:
            sys.exit(f"check_wide_isa: expected one gfx950 code object in {obj}, found {cos}")
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, cos[0])], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def functions(asm, needle):
    """{mangled name: [(address, mnemonic, operands, branch target offset)]} of the kernels whose name contains `needle`"""
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
        if m:
            cur = out.setdefault(m.group(1), []) if needle in m.group(1) else None
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", line)
        if m:
            t = re.search(r"<[^>+]+\+0x([0-9a-f]+)>", m.group(4))  # a branch's target, as objdump resolves it
            cur.append((int(m.group(3), 16), m.group(1), m.group(2), int(t.group(1), 16) if t else None))
    return out


def check(name, insts, verbose=False):
    """-> list of violations"""
    base = insts[0][0]
    index_of = {ins[0]: i for i, ins in enumerate(insts)}
    bad = []
    fifo = []  # outstanding vector-memory operations, oldest first: (index, set of destination registers)
    stats = dict(loads=0, stores=0, waits=0, loops=0, max_outstanding=0)

    def step(i, replay):
        addr, mn, ops, _ = insts[i]
        if FORBIDDEN.match(mn):
            bad.append(f"+0x{addr - base:x}: {mn} {ops}: a memory operation the hand count does not cover")
        pending = set().union(*(d for _, d in fifo)) if fifo else set()
        is_load = bool(VMEM_LOAD.match(mn))
        first, _, rest = ops.partition(",")
        touched = regs(rest if is_load else ops)
        dest = regs(first) if is_load else set()
        hit = (touched | dest) & pending
        if hit and mn != "s_waitcnt":
            owner = [insts[j][0] - base for j, d in fifo if d & hit]
            bad.append(f"+0x{addr - base:x}: {mn} {ops}: touches {sorted(hit)[:4]} while the load(s) at "
                       f"{['+0x%x' % o for o in owner[:3]]} may still be in flight" + (" (second pass over the loop)" if replay else ""))
        if is_load:
            fifo.append((i, dest))
            stats["loads"] += not replay
        elif VMEM_OTHER.match(mn):
            fifo.append((i, set()))
            stats["stores"] += not replay
        elif mn == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ops)
            if m:
                n = int(m.group(1))
                del fifo[:max(0, len(fifo) - n)]
                stats["waits"] += not replay
        stats["max_outstanding"] = max(stats["max_outstanding"], len(fifo))

    i = 0
    while i < len(insts):
        addr, mn, ops, toff = insts[i]
        step(i, False)
        if (mn.startswith("s_cbranch") or mn == "s_branch") and toff is not None:
            target = base + toff
            if target <= addr and target in index_of:
                stats["loops"] += 1
                for j in range(index_of[target], i + 1):  # once more, from the state the back edge brings
                    step(j, True)
        i += 1
    if verbose or bad:
        print(f"{name}: {len(insts)} instructions, {stats['loads']} loads, {stats['stores']} stores/atomics, {stats['waits']} counted waits, "
              f"{stats['loops']} loop(s), at most {stats['max_outstanding']} operations in flight: {'UNSOUND' if bad else 'sound'}")
    return bad, stats


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    needle = "k_partials_mfma_wide"
    if "--kernel" in sys.argv:
        needle = sys.argv[sys.argv.index("--kernel") + 1]
        args = [a for a in args if a != needle]
    verbose = "--verbose" in sys.argv
    if len(args) != 1:
        sys.exit(__doc__)
    fns = functions(disassemble(args[0]), needle)
    if not fns:
        sys.exit(f"check_wide_isa: no kernel named *{needle}* in {args[0]}")
    failed = 0
    for name, insts in sorted(fns.items()):
        bad, stats = check(name, insts, verbose)
        if stats["loads"] == 0 or stats["waits"] == 0 or stats["loops"] == 0:
            bad.append("no loads / counted waits / loop found: the disassembly was not understood")
        for b in bad[:20]:
            print(f"  {name} {b}", file=sys.stderr)
        failed += bool(bad)
    if failed:
        sys.exit(f"check_wide_isa: {failed} of {len(fns)} instantiation(s) of {needle} are not sound as compiled - "
                 "a load's destination is touched before the wait that retires it (see above)")
    print(f"check_wide_isa: {len(fns)} instantiation(s) of {needle} sound as compiled")


if __name__ == "__main__":
    main()
