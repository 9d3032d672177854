"""The C ABI is usable from a host with no Python/torch in the process: compile tests/cabi/cabi_smoke.c
with gcc against include/hpcla_rocm.h + libhpcla_rocm.so and run it (the Julia extension's situation)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi", "cabi_smoke.c")
EXE = os.path.join(ROOT, "tests", "cabi", "_build", "cabi_smoke")
LIBDIR = os.path.join(ROOT, "linearalgebrampi.jl_amd")


SRC_PAIR = os.path.join(ROOT, "tests", "cabi", "cabi_window_pair.c")
EXE_PAIR = os.path.join(ROOT, "tests", "cabi", "_build", "cabi_window_pair")


SRC_RANKS = os.path.join(ROOT, "tests", "cabi", "cabi_ranks_threads.c")
EXE_RANKS = os.path.join(ROOT, "tests", "cabi", "_build", "cabi_ranks_threads")


def _build(src=SRC, exe=EXE):
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    # plain gcc: the header must be valid C; HIP enters only as the runtime API for hipMalloc/hipMemcpy
    subprocess.check_call(["gcc", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-O2", "-Wall",
                           "-Werror=implicit-function-declaration",
                           "-D__HIP_PLATFORM_AMD__", src, "-I", os.path.join(ROOT, "include"),
                           "-I", "/opt/rocm/include", "-L", LIBDIR, "-lhpcla_rocm", "-L", "/opt/rocm/lib",
                           "-lamdhip64", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-lm", "-lpthread", "-o", exe])


def test_cabi_program_builds():
    """CPU: the header is valid C and every symbol the program uses links."""
    _build()
    assert os.path.exists(EXE)
    _build(SRC_PAIR, EXE_PAIR)
    assert os.path.exists(EXE_PAIR)
    _build(SRC_RANKS, EXE_RANKS)
    assert os.path.exists(EXE_RANKS)


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [8, 3])
def test_cabi_eight_ranks_seven_neighbours_in_one_process(nranks):
    """The target's shape (BASELINE configs 3-5: 8 ranks; config 5: 7 neighbours each way), which one-process-per-rank
    workers cannot reach on a box that allows 6 GPU processes: 8 ranks as 8 host threads of ONE process on one GPU, each
    with its own communicator, stream and plans (tests/cabi/cabi_ranks_threads.c).  All-to-all plan (7 send + 7 recv
    neighbours, scattered sends) and slab plan (neighbours next door): probe over every ghost slot, 6 fused distributed
    SpMVs and 3 products with k = 16 bit-exact against the stored-order CPU loop, window all-reduce with 8 slots giving
    identical dot bits on all ranks.  3 ranks: the same program at an odd count."""
    if not os.path.exists(EXE_RANKS):
        _build(SRC_RANKS, EXE_RANKS)
    # (GPU_MAX_HW_QUEUES in the ENVIRONMENT, not only by the program's own setenv: the HIP runtime may read its flags when the
    #  library's code object registers itself, before main(); every rank's streams need hardware queues of their own)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES", "32"))
    out = subprocess.run([EXE_RANKS, str(nranks)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert f"C-ABI {nranks} ranks in one process PASS" in out.stdout
    assert "alltoall:" in out.stdout and "slab:" in out.stdout


@pytest.mark.gpu
def test_cabi_two_processes_push_transport_without_python():
    """Two plain-C processes (fork before any HIP call) bootstrap the peer windows over a socket pair -- the place
    of MPI.Allgather in the Julia extension -- and run distributed SpMVs + window all-reduces through the header
    alone: bit-exact against the closed form, identical dot bits on both ranks."""
    if not os.path.exists(EXE_PAIR):
        _build(SRC_PAIR, EXE_PAIR)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([EXE_PAIR], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "C-ABI window pair PASS" in out.stdout


@pytest.mark.gpu
def test_cabi_program_runs_without_python_runtime():
    if not os.path.exists(EXE):
        _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "C-ABI smoke PASS" in out.stdout


@pytest.mark.gpu
def test_cabi_program_benchmarks_the_headline_workload_without_python():
    """Same kernel, same workload (4096^2 Poisson), no Python in the process: the rate must reach the
    BASELINE target (>= 0.60 of the 8 TB/s HBM peak by algorithmic bytes)."""
    if not os.path.exists(EXE):
        _build()
    out = subprocess.run([EXE, "4096"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    frac = float(out.stdout.strip().split("=")[-1].split("of")[0])
    assert frac >= 0.60, out.stdout
