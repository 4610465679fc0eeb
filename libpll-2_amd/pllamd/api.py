"""ctypes binding of the libpll-2 C ABI for the hot path (struct layouts: SURVEY.md section 8b,
reference src/pll.h:241-335).

The binding is ABI-level, so the same `PllLib` class drives either libpll_amd.so (the product)
or any other library exporting the same symbols; parity tests use that to issue the identical
call sequence to both sides. This module is plumbing for tests and bench.py; the product itself
is the C/HIP library.
"""
import ctypes as C
import os

import numpy as np

c_uint_p = C.POINTER(C.c_uint)
c_double_p = C.POINTER(C.c_double)
c_ubyte_p = C.POINTER(C.c_ubyte)
c_state_p = C.POINTER(C.c_ulonglong)

# attribute word (include/pll_amd.h)
ARCH_CPU, ARCH_SSE, ARCH_AVX, ARCH_AVX2 = 0, 1, 2, 4
PATTERN_TIP = 1 << 4
AB_LEWIS, AB_FELSENSTEIN, AB_STAMATAKIS = 1 << 5, 2 << 5, 3 << 5
AB_FLAG = 1 << 8
RATE_SCALERS = 1 << 9
SITE_REPEATS = 1 << 10
SCALE_BUFFER_NONE = -1

DIRTY_PMATRIX, DIRTY_FREQS, DIRTY_RATE_WEIGHTS, DIRTY_PATTERN_WEIGHTS = 1, 2, 4, 8
DIRTY_INVARIANT, DIRTY_CLV, DIRTY_SCALER, DIRTY_TIPCHARS, DIRTY_REPEATS = 16, 32, 64, 128, 256
DIRTY_EIGEN = 512
FORGET_REPEATS = 1024


class Repeats(C.Structure):
    _fields_ = [
        ("pernode_site_id", C.POINTER(c_uint_p)),
        ("pernode_id_site", C.POINTER(c_uint_p)),
        ("pernode_ids", c_uint_p),
        ("perscale_ids", c_uint_p),
        ("pernode_allocated_clvs", c_uint_p),
        ("enable_repeats", C.c_void_p),
        ("reallocate_repeats", C.c_void_p),
        ("lookup_buffer", c_uint_p),
        ("toclean_buffer", c_uint_p),
        ("id_site_buffer", c_uint_p),
        ("bclv_buffer", c_double_p),
        ("lookup_buffer_size", C.c_uint),
        ("charmap", C.c_char_p),
    ]


class Partition(C.Structure):
    _fields_ = [
        ("tips", C.c_uint),
        ("clv_buffers", C.c_uint),
        ("nodes", C.c_uint),
        ("states", C.c_uint),
        ("sites", C.c_uint),
        ("pattern_weight_sum", C.c_uint),
        ("rate_matrices", C.c_uint),
        ("prob_matrices", C.c_uint),
        ("rate_cats", C.c_uint),
        ("scale_buffers", C.c_uint),
        ("attributes", C.c_uint),
        ("alignment", C.c_size_t),
        ("states_padded", C.c_uint),
        ("clv", C.POINTER(c_double_p)),
        ("pmatrix", C.POINTER(c_double_p)),
        ("rates", c_double_p),
        ("rate_weights", c_double_p),
        ("subst_params", C.POINTER(c_double_p)),
        ("scale_buffer", C.POINTER(c_uint_p)),
        ("frequencies", C.POINTER(c_double_p)),
        ("prop_invar", c_double_p),
        ("invariant", C.POINTER(C.c_int)),
        ("pattern_weights", c_uint_p),
        ("eigen_decomp_valid", C.POINTER(C.c_int)),
        ("eigenvecs", C.POINTER(c_double_p)),
        ("inv_eigenvecs", C.POINTER(c_double_p)),
        ("eigenvals", C.POINTER(c_double_p)),
        ("maxstates", C.c_uint),
        ("tipchars", C.POINTER(c_ubyte_p)),
        ("charmap", c_ubyte_p),
        ("ttlookup", c_double_p),
        ("tipmap", c_state_p),
        ("asc_bias_alloc", C.c_int),
        ("asc_additional_sites", C.c_int),
        ("repeats", C.POINTER(Repeats)),
    ]


class Operation(C.Structure):
    _fields_ = [
        ("parent_clv_index", C.c_uint),
        ("parent_scaler_index", C.c_int),
        ("child1_clv_index", C.c_uint),
        ("child1_matrix_index", C.c_uint),
        ("child1_scaler_index", C.c_int),
        ("child2_clv_index", C.c_uint),
        ("child2_matrix_index", C.c_uint),
        ("child2_scaler_index", C.c_int),
    ]


class Msa(C.Structure):
    """pll_msa_t (src/pll.h:347-354)"""
    _fields_ = [("count", C.c_int), ("length", C.c_int), ("sequence", C.POINTER(C.c_void_p)), ("label", C.POINTER(C.c_void_p))]


assert C.sizeof(Partition) == 232 and C.sizeof(Repeats) == 104 and C.sizeof(Operation) == 32

PartitionP = C.POINTER(Partition)

# symbol -> (restype, argtypes); the hot path and what feeds it
_PROTOS = {
    "pll_partition_create": (PartitionP, [C.c_uint] * 9),
    "pll_partition_destroy": (None, [PartitionP]),
    "pll_set_tip_states": (C.c_int, [PartitionP, C.c_uint, c_state_p, C.c_char_p]),
    "pll_set_tip_clv": (C.c_int, [PartitionP, C.c_uint, c_double_p, C.c_int]),
    "pll_set_pattern_weights": (None, [PartitionP, c_uint_p]),
    "pll_set_asc_bias_type": (C.c_int, [PartitionP, C.c_int]),
    "pll_set_asc_state_weights": (None, [PartitionP, c_uint_p]),
    "pll_set_frequencies": (None, [PartitionP, C.c_uint, c_double_p]),
    "pll_set_subst_params": (None, [PartitionP, C.c_uint, c_double_p]),
    "pll_set_category_rates": (None, [PartitionP, c_double_p]),
    "pll_set_category_weights": (None, [PartitionP, c_double_p]),
    "pll_update_invariant_sites_proportion": (C.c_int, [PartitionP, C.c_uint, C.c_double]),
    "pll_update_invariant_sites": (C.c_int, [PartitionP]),
    "pll_count_invariant_sites": (C.c_uint, [PartitionP, c_uint_p]),
    "pll_update_eigen": (C.c_int, [PartitionP, C.c_uint]),
    "pll_update_prob_matrices": (C.c_int, [PartitionP, c_uint_p, c_uint_p, c_double_p, C.c_uint]),
    "pll_compute_gamma_cats": (C.c_int, [C.c_double, C.c_uint, c_double_p, C.c_int]),
    "pll_update_partials": (None, [PartitionP, C.POINTER(Operation), C.c_uint]),
    "pll_update_partials_rep": (None, [PartitionP, C.POINTER(Operation), C.c_uint, C.c_uint]),
    "pll_compute_edge_loglikelihood": (
        C.c_double,
        [PartitionP, C.c_uint, C.c_int, C.c_uint, C.c_int, C.c_uint, c_uint_p, c_double_p],
    ),
    "pll_compute_root_loglikelihood": (
        C.c_double,
        [PartitionP, C.c_uint, C.c_int, c_uint_p, c_double_p],
    ),
    "pll_update_sumtable": (C.c_int, [PartitionP, C.c_uint, C.c_uint, C.c_int, C.c_int, c_uint_p, c_double_p]),
    "pll_compute_likelihood_derivatives": (
        C.c_int,
        [PartitionP, C.c_int, C.c_int, C.c_double, c_uint_p, c_double_p, c_double_p, c_double_p],
    ),
    "pll_repeats_enabled": (C.c_int, [PartitionP]),
    "pll_get_sites_number": (C.c_uint, [PartitionP, C.c_uint]),
    "pll_get_clv_size": (C.c_uint, [PartitionP, C.c_uint]),
    "pll_get_site_id": (c_uint_p, [PartitionP, C.c_uint]),
    "pll_get_id_site": (c_uint_p, [PartitionP, C.c_uint]),
    "pll_update_repeats": (None, [PartitionP, C.POINTER(Operation)]),
    "pll_disable_bclv": (None, [PartitionP]),
    "pll_resize_repeats_lookup": (None, [PartitionP, C.c_uint]),
    "pll_compress_site_patterns": (c_uint_p, [C.POINTER(C.c_void_p), c_state_p, C.c_int, C.POINTER(C.c_int)]),
    "pll_compress_site_patterns_msa": (c_uint_p, [C.c_void_p, c_state_p, c_uint_p]),
}

# device-residency extension of libpll_amd.so (include/pll_amd.h); absent from other libraries
_GPU_PROTOS = {
    "pll_gpu_sync_clv": (C.c_int, [PartitionP, C.c_uint]),
    "pll_gpu_sync_scaler": (C.c_int, [PartitionP, C.c_uint]),
    "pll_gpu_sync_pmatrix": (C.c_int, [PartitionP, C.c_int]),
    "pll_gpu_edge_loglikelihood_async": (C.c_int, [PartitionP, C.c_uint, C.c_int, C.c_uint, C.c_int, C.c_uint, c_uint_p, C.c_void_p]),
    "pll_gpu_last_algorithmic_bytes": (C.c_double, [PartitionP]),
    "pll_gpu_sync_repeats": (C.c_int, [PartitionP, C.c_int]),
    "pll_gpu_sync_all": (C.c_int, [PartitionP]),
    "pll_gpu_invalidate": (None, [PartitionP, C.c_uint, C.c_int]),
    "pll_gpu_sync_sumtable": (C.c_int, [PartitionP, c_double_p]),
    "pll_gpu_release_sumtable": (C.c_int, [PartitionP, c_double_p]),
    "pll_gpu_set_stream": (C.c_int, [PartitionP, C.c_void_p]),
    "pll_gpu_get_stream": (C.c_void_p, [PartitionP]),
    "pll_gpu_synchronize": (C.c_int, [PartitionP]),
    "pll_gpu_timer_start": (C.c_int, [PartitionP]),
    "pll_gpu_timer_stop": (C.c_double, [PartitionP]),
    "pll_gpu_last_launch_count": (C.c_uint, [PartitionP]),
    "pll_gpu_last_update_replayed": (C.c_int, [PartitionP]),
    "pll_gpu_class_map_work": (C.c_ulonglong, [PartitionP, C.c_int]),
    "pll_gpu_plan_replays": (C.c_ulonglong, [PartitionP]),
    "pll_core_seam_release": (None, []),
    "pll_gpu_group_join": (C.c_void_p, [C.c_char_p, C.c_uint, C.c_uint, C.c_int]),
    "pll_gpu_group_leave": (None, [C.c_void_p]),
    "pll_gpu_group_rank": (C.c_uint, [C.c_void_p]),
    "pll_gpu_group_size": (C.c_uint, [C.c_void_p]),
    "pll_gpu_group_sum": (C.c_int, [C.c_void_p, c_double_p, C.c_uint, c_double_p]),
    "pll_gpu_group_edge_loglikelihood": (
        C.c_double, [PartitionP, C.c_void_p, C.c_uint, C.c_int, C.c_uint, C.c_int, C.c_uint, c_uint_p, c_double_p]),
    "pll_gpu_group_likelihood_derivatives": (
        C.c_int, [PartitionP, C.c_void_p, C.c_int, C.c_int, C.c_double, c_uint_p, c_double_p, c_double_p, c_double_p]),
    "pll_gpu_allreduce_lnl": (C.c_int, [PartitionP, C.c_void_p, C.c_void_p, C.c_uint]),
    "pll_gpu_allreduce_prepare": (C.c_int, [PartitionP, C.c_void_p]),
    "pll_gpu_edge_loglikelihood_allreduce": (
        C.c_double, [PartitionP, C.c_void_p, C.c_uint, C.c_int, C.c_uint, C.c_int, C.c_uint, c_uint_p]),
    "pll_gpu_rccl_available": (C.c_int, []),
    "pll_gpu_device_count": (C.c_int, []),
    "pll_gpu_available": (C.c_int, []),
}


def default_library_path():
    if os.environ.get("PLL_AMD_LIB"):  # A/B builds of the same library (tools/, not the product default)
        return os.environ["PLL_AMD_LIB"]
    here = os.path.dirname(os.path.abspath(__file__))
    return os.path.join(os.path.dirname(here), "csrc", "libpll_amd.so")


class PllLib:
    """One loaded library with the libpll ABI."""

    def __init__(self, path=None):
        self.path = path or default_library_path()
        if not os.path.exists(self.path):
            raise FileNotFoundError(
                f"{self.path} not found - build it first (python -c 'import __graft_entry__ as g; g.build()')"
            )
        # RTLD_LOCAL + DEEPBIND: several libraries exporting the same pll_* names can coexist in one
        # process (each binds to its own definitions). LAZY: a library built without the newick
        # parsers leaves symbols undefined that the hot path never calls.
        mode = getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_LAZY", 1)
        if not os.environ.get("PLL_AMD_NO_DEEPBIND"):  # sanitizer runtimes refuse DEEPBIND (tools/host_asan.sh)
            mode |= getattr(os, "RTLD_DEEPBIND", 0)
        self.dll = C.CDLL(self.path, mode=mode)
        for table in (_PROTOS, _GPU_PROTOS):
            for name, (res, args) in table.items():
                try:
                    fn = getattr(self.dll, name)
                except AttributeError:
                    continue
                fn.restype = res
                fn.argtypes = args
                setattr(self, name, fn)
        self.is_amd = hasattr(self, "pll_gpu_sync_all")

    # ---- globals -----------------------------------------------------------------------
    def state_map(self, name):
        """pll_map_nt / pll_map_aa / pll_map_bin as a ctypes array of 256 state masks."""
        return (C.c_ulonglong * 256).in_dll(self.dll, name)

    def const_doubles(self, name, n):
        return np.array((C.c_double * n).in_dll(self.dll, name), dtype=np.float64)

    def errno(self):
        return C.c_int.in_dll(self.dll, "pll_errno").value

    def errmsg(self):
        return (C.c_char * 200).in_dll(self.dll, "pll_errmsg").value.decode(errors="replace")


def make_ops(rows):
    """rows: iterable of 8-tuples in pll_operation_t field order."""
    rows = list(rows)
    arr = (Operation * max(len(rows), 1))()
    for o, r in zip(arr, rows):
        (o.parent_clv_index, o.parent_scaler_index, o.child1_clv_index, o.child1_matrix_index,
         o.child1_scaler_index, o.child2_clv_index, o.child2_matrix_index, o.child2_scaler_index) = [int(x) for x in r]
    return arr


def as_np(ptr, n, dtype):
    """View n elements behind a ctypes pointer as a numpy array (no copy)."""
    if not ptr:
        return None
    ctype = {np.float64: C.c_double, np.uint32: C.c_uint, np.int32: C.c_int, np.uint8: C.c_ubyte,
             np.uint64: C.c_ulonglong}[dtype]
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(int(n),))


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def uptr(a):
    return a.ctypes.data_as(c_uint_p)
