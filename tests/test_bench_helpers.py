"""CPU tests of bench.py's host-side helpers (the benchmark itself needs an MI355X)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cpu_baseline_leg_reports_the_contract_fields(orc):
    import bench
    rows = orc.poisson2d_rows(64, 64, 0, 64 * 64)
    ci, cv = orc.compress_columns(rows)
    x = orc.fill_uniform(0, 64 * 64, 1)
    out = bench.cpu_baseline_spmv(rows.rowptr.astype(np.int32), cv.astype(np.int32), rows.vals, x, 0.4)
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in out
    assert out["kind"] == "port" and out["unit"] == "GFLOP/s" and out["cores"] >= 1 and out["value"] > 0
    assert out["scipy_bits_equal_oracle"] is True
    json.dumps(out)                                             # serialisable as part of the JSON line


def test_stdout_guard_keeps_native_prints_off_stdout():
    """Whatever writes to file descriptor 1 while the guard is active (RCCL prints a banner there) must land
    on stderr; what bench.py prints afterwards must be the only stdout line."""
    code = (
        "import os, sys; sys.path.insert(0, %r); import bench\n"
        "with bench._StdoutToStderr():\n"
        "    os.write(1, b'native banner\\n'); print('python chatter')\n"
        "print('{\"the\": \"line\"}')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout == '{"the": "line"}\n'
    assert "native banner" in out.stderr and "python chatter" in out.stderr


def test_usable_cores_respects_override(monkeypatch):
    import bench
    monkeypatch.setenv("HPCLA_CPU_THREADS", "3")
    assert bench.usable_cores() == 3
    monkeypatch.delenv("HPCLA_CPU_THREADS")
    assert 1 <= bench.usable_cores() <= 64


def test_guarded_breakdown_watchdog_prints_the_result_and_exits():
    """The optional transport comparison runs under a watchdog: if it hangs, the finished result line is still
    printed (once, on the real stdout) and the process exits with the verification's code."""
    import subprocess
    code = ("import os, sys, time, json\n"
            f"sys.path.insert(0, {ROOT!r})\n"
            "import bench\n"
            "res = {'metric': 'm', 'value': 1.0}\n"
            "bench._guarded_breakdown(lambda: time.sleep(30), res, True, 1)\n"
            "print('not reached')\n")
    env = dict(os.environ, HPCLA_BENCH_BREAKDOWN_TIMEOUT_S="0.3")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, env=env)
    assert out.returncode == 0, out.stderr[-500:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and "not reached" not in out.stdout
    rec = json.loads(lines[0])
    assert rec["value"] == 1.0 and "error" in rec["step_breakdown_ms_max_over_ranks"]
    # and the ordinary case: the comparison's record comes back, nothing is printed by the guard
    code2 = ("import sys\n"
             f"sys.path.insert(0, {ROOT!r})\n"
             "import bench\n"
             "print(bench._guarded_breakdown(lambda: {'modes': {}}, {'value': 2.0}, True, 1))\n")
    out2 = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=60)
    assert out2.returncode == 0 and out2.stdout.strip() == "{'modes': {}}", (out2.stdout, out2.stderr[-300:])
