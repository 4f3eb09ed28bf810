import importlib.util
import json
import os
import sys

import numpy as np
import pytest

try:  # torch first: it brings its own HIP runtime, and the library must share that one instance
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "ceres-solver-ceres-solver_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _load_package():
    """The package directory name has hyphens; load it as module `cxschur`."""
    if "cxschur" in sys.modules:
        return sys.modules["cxschur"]
    spec = importlib.util.spec_from_file_location(
        "cxschur", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cxschur"] = mod
    spec.loader.exec_module(mod)
    return mod


def _load_oracle():
    if "orc" in sys.modules:
        return sys.modules["orc"]
    spec = importlib.util.spec_from_file_location("orc", os.path.join(ROOT, "oracle", "orc.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["orc"] = mod
    spec.loader.exec_module(mod)
    return mod


cx = _load_package()
orc = _load_oracle()


@pytest.fixture(scope="session")
def cxschur():
    return cx


@pytest.fixture(scope="session")
def oracle():
    orc.lib()
    orc.set_num_threads(4)
    return orc


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def lls_problem(pid):
    """(BlockStructure, values, b, D, num_eliminate_blocks, raw dict) of fixture problem `pid`."""
    raw = load_golden("linear_least_squares_problems.json")["problems"][str(pid)]
    rows = [(rs, [tuple(c) for c in cells]) for rs, cells in raw["rows"]]
    bs = cx.BlockStructure.from_rows(raw["col_sizes"], rows)
    return (bs, np.array(raw["values"], dtype=np.float64), np.array(raw["b"], dtype=np.float64),
            np.array(raw["D"], dtype=np.float64), raw["num_eliminate_blocks"], raw)
