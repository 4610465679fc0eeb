"""Python plumbing (ctypes) around libpll_amd.so, the MI355X build of the libpll-2
partial-likelihood hot path. The product is the C/HIP library under ../csrc; this package only
binds its C ABI for tests and bench.py."""
from . import api, driver, sharding, workload  # noqa: F401
from .api import PllLib  # noqa: F401
