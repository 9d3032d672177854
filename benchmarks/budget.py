"""Time budget of one bench.py run (pure Python, no GPU: tests/test_bench_helpers.py covers it).

The driver gives a bench run an outer wall-clock limit (600 s in round 2's records).  Everything optional in
bench.py -- the sub-records beside the headline, the transport comparison -- and every guard around an untested
leg (RCCL bootstrap, the comparison's watchdog, the launcher's own limit) is derived from that ONE number
(HPCLA_BENCH_OUTER_LIMIT_S, default 600), so the guards can never add up to more than the limit and an
optional record is skipped ("skipped": "budget") rather than started when the time left is short.

The origin of the clock is the start of the PARENT process (`python bench.py --gpus N` spawns its ranks: the
spawn, the imports and the rendezvous are inside the driver's limit too); it travels to the ranks in
HPCLA_BENCH_T0 (epoch seconds).
"""
import os
import sys
import time

RESERVE_S = 30.0              # kept back for teardown, process exit and the driver's own bookkeeping

# what an optional stage is assumed to cost (seconds, generous: measured 2-15 s each on one GPU; a stage is only
# started when the remaining budget covers its estimate)
ESTIMATE_S = {
    "strong_scaling": 40.0,           # the fixed 8192^2 problem over all ranks
    "strong_scaling_n1": 35.0,        # the same problem on rank 0 alone (gives speedup_vs_n1)
    "strong_scaling_4096": 15.0,      # the 4096^2 headline matrix strong-scaled (the other reading of ">= 6x at 8 GPUs")
    "strong_scaling_4096_n1": 12.0,   # ... and on rank 0 alone
    "int64": 20.0,
    "float32": 15.0,                  # the headline matrix on a Float32 backend (csrc/f32.hip)
    "poisson3d_cg": 30.0,
    "poisson2d_spmm": 20.0,
    "sprand_spmm": 35.0,
    "sprand_spmm_mall_sized": 25.0,
    "sprand_spmm_panel_order": 45.0,  # N > 1 only: four more chunk-set plans to attach
    "packed": 15.0,
    "cpu_baseline": 30.0,
    "comparison": 45.0,               # minimum worth starting; its watchdog gets what is left, <= 90 s
}


class Budget:
    def __init__(self, outer_s=None, t0=None, clock=time.time, out=None):
        if outer_s is None:
            outer_s = float(os.environ.get("HPCLA_BENCH_OUTER_LIMIT_S", "600"))
        self.outer = float(outer_s)
        self.clock = clock
        if t0 is None:
            t0 = float(os.environ.get("HPCLA_BENCH_T0", "0") or 0) or clock()
        self.t0 = float(t0)
        self.out = out if out is not None else sys.stderr
        self.skipped = []

    # ---- the clock ---------------------------------------------------------------------------------
    def elapsed(self) -> float:
        return self.clock() - self.t0

    def remaining(self) -> float:
        """Seconds that may still be spent on measurements (the reserve is not available)."""
        return self.outer - RESERVE_S - self.elapsed()

    # ---- guards derived from the outer limit -----------------------------------------------------------
    def launcher_timeout(self) -> float:
        """The self-launching parent kills its ranks this long after ITS start: outer - 30 s, never the other way
        round (round 2's launcher allowed 1500 s inside a 600 s driver limit)."""
        return max(self.outer - RESERVE_S, 30.0)

    def rccl_init_timeout(self) -> float:
        return max(10.0, min(60.0, 0.1 * self.outer))

    def comparison_timeout(self) -> float:
        """Watchdog of the transport comparison: what is left, at most 90 s."""
        return max(0.0, min(90.0, self.remaining()))

    def spin_timeout(self) -> float:
        """In-kernel spin bound of the push transport during a bench run (library default: 20 s)."""
        return max(5.0, min(20.0, 0.03 * self.outer))

    def export_guards(self, env=os.environ) -> None:
        """Defaults for the library's own guards (explicit settings win)."""
        env.setdefault("HPCLA_RCCL_INIT_TIMEOUT_S", f"{self.rccl_init_timeout():.0f}")
        env.setdefault("HPCLA_PUSH_TIMEOUT_S", f"{self.spin_timeout():.0f}")
        env.setdefault("HPCLA_BENCH_T0", repr(self.t0))

    # ---- optional stages -----------------------------------------------------------------------------
    def allows(self, name: str, need_s=None) -> bool:
        need = ESTIMATE_S.get(name, 30.0) if need_s is None else float(need_s)
        ok = self.remaining() >= need
        if not ok:
            self.skipped.append(name)
            self.stage(f"SKIP {name}: needs ~{need:.0f} s, {max(self.remaining(), 0):.0f} s left of {self.outer:.0f}")
        return ok

    def stage(self, name: str) -> None:
        self.out.write(f"[bench +{self.elapsed():7.1f}s] {name}\n")
        try:
            self.out.flush()
        except Exception:
            pass


SKIPPED = {"skipped": "budget"}
