"""The C ABI is usable from a host with no Python/torch in the process: compile tests/cabi/cabi_smoke.c
with gcc against include/hpcla_rocm.h + libhpcla_rocm.so and run it (the Julia extension's situation)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi", "cabi_smoke.c")
EXE = os.path.join(ROOT, "tests", "cabi", "_build", "cabi_smoke")
LIBDIR = os.path.join(ROOT, "linearalgebrampi.jl_amd")


def _build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    # plain gcc: the header must be valid C; HIP enters only as the runtime API for hipMalloc/hipMemcpy
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Werror=implicit-function-declaration",
                           "-D__HIP_PLATFORM_AMD__", SRC, "-I", os.path.join(ROOT, "include"),
                           "-I", "/opt/rocm/include", "-L", LIBDIR, "-lhpcla_rocm", "-L", "/opt/rocm/lib",
                           "-lamdhip64", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", EXE])


def test_cabi_program_builds():
    """CPU: the header is valid C and every symbol the program uses links."""
    _build()
    assert os.path.exists(EXE)


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
