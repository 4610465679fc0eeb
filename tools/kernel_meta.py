#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel in a built object, from the code object's metadata notes (no -save-temps
needed): tools/kernel_meta.py libpll-2_amd/csrc/hip/pllgpu.o [substring]. vgpr = architected + accumulation registers
the wave occupies (what decides waves per SIMD on gfx950: 512 / vgpr, DESIGN.md "Registers")."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def main():
    obj, needle = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    tmp = tempfile.mkdtemp(prefix="kmeta_")
    try:
        local = os.path.join(tmp, "unit.o")
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True)
        co = [f for f in os.listdir(tmp) if "gfx950" in f][0]
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    rows = []
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        if needle not in name:
            continue
        agpr = blk.split("\n")[0].strip()
        rows.append((name, g("vgpr_count"), agpr, g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("max_flat_workgroup_size")))
    dem = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.splitlines()
    for r, d in zip(rows, dem):
        d = re.sub(r"^void ", "", d)
        d = re.sub(r"\(.*$", "", d)
        v = int(r[1]) if r[1].isdigit() else 0
        waves = 512 // max(8, (v + 7) // 8 * 8) if v else 0
        print(f"{d[:70]:70s} vgpr={r[1]:>4} (agpr {r[2]:>3}) sgpr={r[3]:>4} scratch={r[4]:>5} lds={r[5]:>6} wg={r[6]:>5} waves/SIMD<={min(waves, 8)}")


if __name__ == "__main__":
    main()
