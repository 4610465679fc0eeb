"""GPU: a C program compiled against include/pll_amd.h and linked with libpll_amd.so - the drop-in
boundary exercised the way a C application would, not through ctypes."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_caller_reproduces_the_reference_kat(tmp_path):
    exe = str(tmp_path / "dropin")
    libdir = os.path.join(ROOT, "libpll-2_amd", "csrc")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_caller", "dropin.c"),
                           "-L" + libdir, "-lpll_amd", "-lm", "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines())
    assert abs(float(lines["lnl"].split()[0]) - (-58.887310)) < 5.1e-7


def _build_sharded(tmp_path):
    exe = str(tmp_path / "sharded")
    libdir = os.path.join(ROOT, "libpll-2_amd", "csrc")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "c_caller"),
                           os.path.join(ROOT, "tests", "c_caller", "sharded.c"),
                           "-L" + libdir, "-lpll_amd", "-ldl", "-lm", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


@pytest.mark.parametrize("world", [1, 2, 3])
def test_c_caller_sharded_over_the_fixed_order_exchange(tmp_path, world):
    """the 12-site KAT cut into `world` ranges, one process and partition per range, summed by
    pll_gpu_group_edge_loglikelihood: every rank reports the reference's value, the same bits in all 50 steps"""
    out = subprocess.run([_build_sharded(tmp_path), "peer", str(world)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    vals = [ln.split()[-1] for ln in out.stdout.strip().splitlines()]
    assert len(vals) == world and len(set(vals)) == 1  # the same bits (%a) on every rank


def test_c_caller_all_reduces_through_rccl_without_python(tmp_path):
    """the north_star's "single RCCL all-reduce" reachable from C - a real communicator of one
    rank handed to pll_gpu_edge_loglikelihood_allreduce (librccl bound by the library with dlopen)"""
    out = subprocess.run([_build_sharded(tmp_path), "rccl"], capture_output=True, text=True, timeout=600)
    if out.returncode == 77:
        pytest.skip("no RCCL library on this host")
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rccl lnl -58.88731" in out.stdout


def _build_rccl_double(tmp_path):
    """tests/c_caller/rccl_double.c: the stream-ordered stand-in for librccl (test infrastructure)"""
    so = str(tmp_path / "librccl_double.so")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "c_caller", "rccl_double.c"), "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath,/opt/rocm/lib", "-o", so])
    return so


@pytest.mark.parametrize("world", [2, 3])
def test_c_caller_collective_evaluation_with_several_ranks(tmp_path, world):
    """pll_gpu_edge_loglikelihood_allreduce had only ever met a one-rank communicator. `world` forked
    ranks on this one device, the communicator a stand-in that keeps RCCL's contract (asynchronous, on the
    partition's stream, device operands): the reduced sequence word is ranks x step, every rank returns the same
    bits = the reference's value for the 12 sites, a step in which the last rank's evaluation fails gives -inf
    everywhere with each rank's own pll_errno, and the next step is in step again"""
    out = subprocess.run([_build_sharded(tmp_path), "double", str(world), _build_rccl_double(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    vals = [ln.split()[-1] for ln in out.stdout.strip().splitlines()]
    assert len(vals) == world and len(set(vals)) == 1


def test_c_caller_collective_evaluation_with_a_missing_rank_returns(tmp_path):
    """a peer that never joins the all-reduce used to leave the others in an unbounded
    hipStreamSynchronize; now they poll the stream for PLL_AMD_REDUCE_TIMEOUT_MS and return -inf + an error"""
    out = subprocess.run([_build_sharded(tmp_path), "missing", "2", _build_rccl_double(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "did not complete" in out.stdout


def _captured(fn):
    """what a C-level printf writes to stdout while fn() runs"""
    import ctypes as C
    import os
    import tempfile
    libc = C.CDLL(None)
    libc.fflush(None)
    with tempfile.TemporaryFile() as tmp:
        saved = os.dup(1)
        os.dup2(tmp.fileno(), 1)
        try:
            fn()
            libc.fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        tmp.seek(0)
        return tmp.read().decode()


@pytest.mark.parametrize("attrs", [0, "repeats"], ids=["plain", "site-repeats"])
def test_show_functions_print_what_the_reference_prints(amd_lib, ref_lib, attrs):
    """pll_show_pmatrix / pll_show_clv (src/output.c): the examples and tests of the reference call
    them; same text for the same partition (6 decimals: far above the 1e-10 the numbers agree to)"""
    import ctypes as C
    from pllamd import api, driver, workload as W
    a = api.SITE_REPEATS if attrs == "repeats" else 0
    case = W.make_case("show", 4, 8, 30, seed=12, attributes=a, mutate_pct=5)
    text = {}
    for lib in (amd_lib, ref_lib):
        lib.dll.pll_show_pmatrix.restype = None
        lib.dll.pll_show_pmatrix.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
        lib.dll.pll_show_clv.restype = None
        lib.dll.pll_show_clv.argtypes = [C.c_void_p, C.c_uint, C.c_int, C.c_uint]
        with driver.Session(lib, case, api.ARCH_AVX2) as s:
            s.update_partials()
            op = case.op_batches[0][-1]

            def show():
                lib.dll.pll_show_pmatrix(s.p, 3, 6)
                lib.dll.pll_show_clv(s.p, op[0], op[1], 6)
                lib.dll.pll_show_clv(s.p, case.op_batches[0][0][0], -1, 6)
            text[lib.is_amd] = _captured(show)
    assert len(text[True]) > 1000
    assert text[True] == text[False]
