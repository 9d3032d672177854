"""CPU tests of the self-launcher (linearalgebrampi.jl_amd/launch.py), the code behind `python bench.py --gpus N`
(the reference's distributed entry launches itself too: test/runtests.jl:16-35), and of the host twin of the
device structure digest."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def _run_launcher(script, nranks, timeout=None):
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import importlib.util as u;"
            f"s = u.spec_from_file_location('l', {os.path.join(ROOT, 'linearalgebrampi.jl_amd', 'launch.py')!r});"
            "m = u.module_from_spec(s); s.loader.exec_module(m);"
            f"sys.exit(m.spawn_ranks([{script!r}], {nranks}, timeout={timeout!r}))")
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)


def test_launcher_is_stdlib_only():
    """The launching parent must never import torch or load the HIP library."""
    src = open(os.path.join(ROOT, "linearalgebrampi.jl_amd", "launch.py")).read()
    assert "import torch" not in src and "hpcla" not in src.split('"""', 2)[2].replace("HPCLA_SELF_LAUNCHED", "")
    out = subprocess.run([sys.executable, "-c",
                          f"import sys, importlib.util as u; s = u.spec_from_file_location('l', "
                          f"{os.path.join(ROOT, 'linearalgebrampi.jl_amd', 'launch.py')!r}); m = u.module_from_spec(s); "
                          "s.loader.exec_module(m); print('torch' in sys.modules)"],
                         capture_output=True, text=True, timeout=60)
    assert out.stdout.strip() == "False"


def test_ranks_get_the_torchrun_environment_and_rank0_owns_stdout(tmp_path):
    script = _script(tmp_path, """
        import os, sys
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1"
        assert int(os.environ["MASTER_PORT"]) > 0
        print(f"line from rank {r} of {w}")
    """)
    out = _run_launcher(script, 3)
    assert out.returncode == 0, out.stderr
    assert out.stdout == "line from rank 0 of 3\n"                      # ONE line: rank 0's
    assert "line from rank 1 of 3" in out.stderr and "line from rank 2 of 3" in out.stderr


def test_a_failing_rank_fails_the_job_and_stops_the_others(tmp_path):
    script = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)          # "blocked in a collective"
    """)
    out = _run_launcher(script, 2)
    assert out.returncode == 7


def test_timeout_stops_a_hung_job(tmp_path):
    script = _script(tmp_path, "import time; time.sleep(60)\n")
    out = _run_launcher(script, 2, timeout=1.5)
    assert out.returncode == 124


def test_free_ports_differ():
    from hpcla_amd.launch import free_port
    assert len({free_port() for _ in range(4)}) >= 2


def test_bench_parent_does_not_import_torch():
    """`python bench.py --gpus 2` on a box without a GPU: the parent launches two ranks, both refuse (no CPU
    fallback), the parent exits non-zero -- and never imported torch itself."""
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--steps', '1'];\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n"
            "    print('RC', e.code, 'TORCH', 'torch' in sys.modules)" % os.path.join(ROOT, "bench.py"))
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""           # also without a GPU when one exists
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert "TORCH False" in out.stdout, out.stdout + out.stderr
    assert "RC 0" not in out.stdout


def test_array_digest_is_order_sensitive_and_width_agnostic():
    from hpcla_amd.partition import array_digest
    a = np.arange(1000, dtype=np.int32)
    assert array_digest(a) == array_digest(a.astype(np.int64))              # Ti is a separate part of the key
    assert array_digest(a) != array_digest(a[::-1].copy())
    b = a.copy()
    b[500], b[501] = b[501], b[500]
    assert array_digest(a) != array_digest(b)
    assert array_digest(np.zeros(5, np.int32)) != array_digest(np.zeros(6, np.int32))
    big = np.arange(5_000_000, dtype=np.int64)                              # crosses the blocking boundary
    assert array_digest(big) == array_digest(big.copy())
    assert len(array_digest(a)) == 32


def test_window_descriptor_helpers_on_the_host():
    """Host-side halves of the window bootstrap that need no GPU: the same-node decision from the gathered
    descriptors (bytes 64..71 = node identity) and the no-op path of attach_halo_windows without windows."""
    import types
    from hpcla_amd import backends as B
    comm = B.CommSerial()
    d = bytes(64) + (1234).to_bytes(8, "little") + bytes(56)
    descs, one = B.allgather_window_descs(comm, d)
    assert descs == [d] and one is True
    fake = types.SimpleNamespace(peer_windows=False, comm=comm, has_rccl=False)
    assert B.attach_halo_windows(fake, None) is False            # nothing to attach, no collective issued
