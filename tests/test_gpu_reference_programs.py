"""The reference's own test programs, run unchanged against libpll_amd.so on the MI355X.

oracle/Makefile (`make reftests`) compiles each data-free program of the reference's test/src
where it lies and links it first against the product library, so every pll_* call it makes -
partition set-up, P-matrices, pll_update_partials, pll_compute_edge_loglikelihood, derivatives,
the printers - is served by the HIP path (the only symbols left to the reference build are the
FASTA reader entry points its common.c mentions and never calls here).  The expected text is the
reference's test/out/<name>.out (test/runtest.py diffs stdout against exactly these files),
committed as fixtures under tests/golden/reference_test_out/.

The attribute sets are runtest.py's twelve (test/runtest.py:47-60).  The reference prints with
five or six decimals; wherever the text differs the test accepts a numeric difference of one
unit in the last printed place and nothing else - with one documented exception: numbers printed
in %e format (the derivatives) may also differ by RESIDUE in absolute terms.  At branch lengths
50 and 90 the true derivative is zero and the reference prints its own summation residue
(-6.6613e-15 = 30 ulp of 1, from per-site terms of magnitude 1..10); a different summation order
leaves a different residue of the same size.
"""
import lzma
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "tests")
OUT = os.path.join(ROOT, "tests", "golden", "reference_test_out")

PROGRAMS = ["00010_NMDU_lkcalc", "00011_NMAU_lkcalc", "00012_NMOU_lkcalc", "00020_NMDR_lkcalc",
            "00021_NMAR_lkcalc", "00022_NMOR_lkcalc", "00030_NMDU_gamma", "00032_NMOU_gamma",
            "alpha-cats", "hky", "derivatives", "derivatives-oddstates", "pmatrix",
            "compress-patterns",
            # no expected output in the reference's test/out: recorded from the reference build
            # itself by `make -C oracle reftests-golden` (same text for every attribute set)
            "protein-models"]
# the self-contained programs of the reference's examples/ (inline data, no arguments, attributes
# fixed in their source); expected text recorded from the reference build, as for protein-models
EXAMPLES = ["example-rooted", "example-unrooted", "example-rooted-tacg", "example-heterotachy"]
ATTRIBUTES = ["", "tv", "avx", "avx tv", "sse", "sse tv", "avx2", "avx2 tv",
              "sr", "avx sr", "sse sr", "avx2 sr"]

RESIDUE = 1e-13

NUMBER = re.compile(r" *([-+]?\d+\.\d+(?:[eE][-+]?\d+)?)")   # padding moves with the sign


def same_text(got, want):
    """Equal, or equal after allowing each printed number one unit in its last place."""
    if got == want:
        return None
    gl, wl = got.splitlines(), want.splitlines()
    if len(gl) != len(wl):
        return f"{len(gl)} lines, expected {len(wl)}"
    for n, (g, w) in enumerate(zip(gl, wl), 1):
        if g == w:
            continue
        if NUMBER.sub("#", g) != NUMBER.sub("#", w):
            return f"line {n}: {g!r} != {w!r}"
        for a, b in zip(NUMBER.findall(g), NUMBER.findall(w)):
            if a == b:
                continue
            if "e" in a.lower() or "e" in b.lower():
                fa, fb = float(a), float(b)
                digits = len(b.lower().split("e")[0].split(".")[1])
                if abs(fa - fb) > max(RESIDUE, 1.5 * 10 ** -digits * max(abs(fa), abs(fb))):
                    return f"line {n}: {a} != {b}"
                continue
            places = len(b.split(".")[1])
            if len(a.split(".")[1]) != places or abs(float(a) - float(b)) > 1.5 * 10 ** -places:
                return f"line {n}: {a} != {b}"
    return None


def expected_output(program):
    path = os.path.join(OUT, program + ".out")
    if os.path.exists(path + ".xz"):
        with lzma.open(path + ".xz", "rt") as f:
            return f.read()
    with open(path) as f:
        return f.read()


@pytest.mark.gpu
@pytest.mark.parametrize("attributes", ATTRIBUTES, ids=[a.replace(" ", "+") or "cpu" for a in ATTRIBUTES])
@pytest.mark.parametrize("program", PROGRAMS)
def test_reference_program_prints_the_expected_output(program, attributes):
    exe = os.path.join(BIN, program)
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not shipped: `make -C oracle reftests` builds it where /root/reference exists")
    run = subprocess.run([exe] + attributes.split(), capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    with open(os.path.join(OUT, "skip.out")) as f:
        if run.stdout == f.read():
            pytest.skip("the program itself skips this attribute set")
    want = expected_output(program)
    problem = same_text(run.stdout, want)
    assert problem is None, problem


@pytest.mark.gpu
@pytest.mark.parametrize("program", EXAMPLES)
def test_reference_example_prints_what_it_prints_with_the_reference(program):
    exe = os.path.join(BIN, program)
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not shipped: `make -C oracle reftests` builds it where /root/reference exists")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    problem = same_text(run.stdout, expected_output(program))
    assert problem is None, problem
