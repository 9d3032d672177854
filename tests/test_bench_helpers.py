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
