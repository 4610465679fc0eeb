"""Run the reference's own test programs (oracle/_ref/tests, linked against libpll_amd.so) under
runtest.py's twelve attribute sets and say, per run, whether stdout is byte-identical to the
reference's expected output or differs within tests/test_gpu_reference_programs.py's tolerance.
Usage (GPU box): python tools/reference_programs_report.py > gpurun_out/reference_programs.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_reference_programs import ATTRIBUTES, BIN, EXAMPLES, OUT, PROGRAMS, expected_output, same_text  # noqa: E402

skip = open(os.path.join(OUT, "skip.out")).read()
tally = {"identical": 0, "within tolerance": 0, "skipped by the program": 0, "DIFFERENT": 0}
print(f"{'program':24s} " + " ".join(f"{(a.replace(' ', '+') or 'cpu'):>8s}" for a in ATTRIBUTES))
for prog in PROGRAMS + EXAMPLES:
    want = expected_output(prog)
    row = []
    for attr in (ATTRIBUTES if prog in PROGRAMS else [""]):
        run = subprocess.run([os.path.join(BIN, prog)] + attr.split(), capture_output=True, text=True)
        if run.stdout == skip:
            key, mark = "skipped by the program", "skip"
        elif run.stdout == want:
            key, mark = "identical", "=="
        elif run.returncode == 0 and same_text(run.stdout, want) is None:
            nd = sum(g != w for g, w in zip(run.stdout.splitlines(), want.splitlines()))
            key, mark = "within tolerance", f"~{nd}"
        else:
            key, mark = "DIFFERENT", "FAIL"
        tally[key] += 1
        row.append(mark)
    print(f"{prog:24s} " + " ".join(f"{m:>8s}" for m in row))
print()
print("== byte-identical stdout; ~N = N lines differ, each within one unit of the last printed place")
print("(or, for %e derivatives whose true value is 0, by summation residue < 1e-13)")
for k, v in tally.items():
    print(f"{k}: {v}")
