"""CPU: tools/check_wide_isa.py - the build-time replay of k_partials_mfma_wide's emitted instruction stream (ADVICE r3:
`wide_kernel_is_sound` only saw scratch spills; a register copy or an accumulation-register spill between an inline-asm
load and its hand-counted s_waitcnt would go unnoticed and give wrong CLVs). The built object must pass, and the replay
must actually catch the failures it is there for: they are planted into the real instruction stream here."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
OBJ = os.path.join(ROOT, "libpll-2_amd", "csrc", "hip", "pllgpu.o")

import check_wide_isa as W  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(OBJ) or not os.path.exists(os.path.join(W.LLVM, "llvm-objdump")),
                                reason="needs the built pllgpu.o and llvm-objdump (the authoring container)")


@pytest.fixture(scope="module")
def kernels():
    fns = W.functions(W.disassemble(OBJ), "k_partials_mfma_wide")
    assert len(fns) == 2  # <15, 1, 8> (61 states, exact) and <16, 0, 8> (padded)
    return fns


def test_the_built_kernels_are_sound(kernels):
    for name, insts in kernels.items():
        bad, stats = W.check(name, insts)
        assert not bad, bad[:3]
        # what the source promises: 16 row requests per child x (first left child + right + next left) + the matrices'
        # staging loads; a loop; dozens of counted waits - i.e. the replay saw the kernel, not an empty function
        assert stats["loads"] >= 48 and stats["waits"] >= 30 and stats["loops"] == 1 and stats["stores"] >= 17


def _first(insts, pred, start=0):
    return next(i for i in range(start, len(insts)) if pred(insts[i]))


def test_a_register_copy_behind_a_load_is_caught(kernels):
    for name, insts in kernels.items():
        i = _first(insts, lambda x: x[1] == "global_load_dwordx4" and " nt" in x[2])
        dest = insts[i][2].split(",")[0].strip()  # v[64:67]
        lo = int(dest[2:].split(":")[0])
        planted = list(insts)
        planted.insert(i + 1, (insts[i][0] + 1, "v_mov_b32_e32", f"v250, v{lo}", None))
        bad, _ = W.check(name, planted)
        assert bad and "v_mov_b32_e32" in bad[0]
        planted[i + 1] = (insts[i][0] + 1, "v_accvgpr_write_b32", f"a3, v{lo + 1}", None)  # a spill to the accumulation file
        bad, _ = W.check(name, planted)
        assert bad and "v_accvgpr_write_b32" in bad[0]


def test_a_wait_that_is_one_short_is_caught(kernels):
    """every counted wait of the loop matters: allowing one more operation to stay in flight lets the MFMAs behind it read a
    row that may not have landed"""
    import re
    for name, insts in kernels.items():
        loop_end = _first(insts, lambda x: x[1].startswith("s_cbranch") and x[3] is not None and x[3] < x[0] - insts[0][0])
        loop_start = next(i for i, x in enumerate(insts) if x[0] - insts[0][0] == insts[loop_end][3])
        waits = [i for i in range(loop_start, loop_end) if insts[i][1] == "s_waitcnt" and re.search(r"vmcnt\((\d+)\)", insts[i][2])]
        assert len(waits) >= 30
        caught = 0
        for i in waits:
            n = int(re.search(r"vmcnt\((\d+)\)", insts[i][2]).group(1))
            planted = list(insts)
            planted[i] = (insts[i][0], "s_waitcnt", re.sub(r"vmcnt\(\d+\)", f"vmcnt({n + 1})", insts[i][2]), None)
            bad, _ = W.check(name, planted)
            caught += bool(bad)
        # (a few waits are followed by another wait before the row is used - the compiler's own for the staging loads -
        # so not every single one is load-bearing; the hand-counted ones are)
        assert caught >= 30, (name, caught, len(waits))


def test_scratch_traffic_is_refused(kernels):
    for name, insts in kernels.items():
        planted = list(insts)
        planted.insert(40, (insts[40][0] + 1, "scratch_store_dword", "off, v3, s32", None))
        bad, _ = W.check(name, planted)
        assert bad and "hand count" in bad[0]
