"""cxschur -- MI355X (gfx950) implementation of Ceres Solver's LM inner linear solve.

The directory name carries hyphens (repo convention), so load this package with
importlib under the module name ``cxschur`` (tests/conftest.py, bench.py and
__graft_entry__.py all do).  The compute path is the C-ABI library
``csrc/libcxschur.so`` (include/cxschur.h); this package is the ctypes binding
used by the tests and the benchmark plus host-side helpers (synthetic BAL
problems, build script).  There is no CPU fallback: every operation raises if
the HIP library is missing or reports an error.
"""
from .structure import BLOCK_DTYPE, CELL_DTYPE, BlockStructure  # noqa: F401
from . import bal, dumps  # noqa: F401

try:  # binding.py needs only ctypes/numpy; the library itself is loaded lazily
    from .binding import *  # noqa: F401,F403
    from . import binding, boundary  # noqa: F401
except ImportError:  # pragma: no cover - only during partial checkouts
    raise
