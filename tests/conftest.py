import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the library's own default spin bound is 300 s (a slow neighbour is waited for like a blocking MPI receive); a test
# run must never sit that long behind a defect
os.environ.setdefault("HPCLA_PUSH_TIMEOUT_S", "30")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the HIP library (hipcc
    cross-compiles gfx950 without a GPU, a few seconds) and the oracle before the first test needs them --
    the same two `make` calls as __graft_entry__.build()."""
    import subprocess
    lib = os.path.join(ROOT, "linearalgebrampi.jl_amd", "libhpcla_rocm.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "linearalgebrampi.jl_amd", "csrc")],
                              stdout=subprocess.DEVNULL)
    built = os.path.join(ROOT, "oracle", "_build")               # where oracle/Makefile puts liborc.so
    if not (os.path.isdir(built) and any(f.endswith(".so") for f in os.listdir(built))):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "hotpath_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def hp():
    import hpcla_amd
    return hpcla_amd


@pytest.fixture(scope="session")
def gpu_backend_i32(hp):
    import numpy as np
    return hp.backend_rocm_serial(np.float64, np.int32)


@pytest.fixture(scope="session")
def gpu_backend_i64(hp):
    import numpy as np
    return hp.backend_rocm_serial(np.float64, np.int64)


@pytest.fixture(scope="session")
def pin_large():
    """tests/golden/pin_large.npz (tests/golden/make_golden_large.py): the reference's own LARGER input definitions --
    laplacian_2d_sparse(10^4), tools/benchmark_vs_petsc.jl:42-49, and a generate_sparse(1000)-shaped matrix,
    tools/benchmark_single_rank.jl:48-71 -- with exact-rational products rounded once.  Data only; loaded with
    numpy's non-executing loader."""
    import numpy as np
    with np.load(os.path.join(ROOT, "tests", "golden", "pin_large.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}
