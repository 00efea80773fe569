"""Exact FP64 operation count of a planner workload (TEST INFRASTRUCTURE ONLY).

oracle/libpmaf_flopcount.so is the CPU restatement (pmaf_oracle.c) compiled as
C++ with `double` replaced by an operation-counting scalar type
(flopcount.cpp): every + - * / sqrt exp and comparison the algorithm executes
is counted as 1 (SURVEY.md 8d's convention). bench.py reports the count per
agent-step of its workload (`flops_per_agent_step_measured`) and the fraction
of agent-steps with at least one obstacle inside the detection shell; nothing
in the product path imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import orc

_HERE = os.path.dirname(os.path.abspath(__file__))


def build():
    so = os.path.join(_HERE, "libpmaf_flopcount.so")
    srcs = [os.path.join(_HERE, f) for f in ("flopcount.cpp", "pmaf_oracle.c", "pmaf_oracle.h", "Makefile")]
    if os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs):
        return so
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libpmaf_flopcount.so"])
    return so


class _CountingOracle(orc.OraclePlanner):
    """OraclePlanner bound to the instrumented library instead of libpmaf_oracle.so"""
    _lib = None

    @classmethod
    def library(cls):
        if cls._lib is None:
            saved, orc._LIB = orc._LIB, None
            os.environ["PMAF_ORACLE_LIB"] = build()
            try:
                cls._lib = orc.lib()     # same prototypes, other shared object
            finally:
                del os.environ["PMAF_ORACLE_LIB"]
                orc._LIB = saved
            for fn in ("orc_flops", "orc_steps_total", "orc_steps_in_shell"):
                getattr(cls._lib, fn).restype = C.c_longlong
        return cls._lib


def count_scene(scene, ticks=24, episode=256):
    """Run `ticks` planner ticks of `scene` (the bench's episode protocol) on the
    instrumented oracle; count only the rollouts (the agent x horizon loop the
    GPU kernel replaces). libm exp mode, so exp counts as ONE operation."""
    L = _CountingOracle.library()
    saved, orc._LIB = orc._LIB, L
    try:
        L.orc_set_exp_mode(0)
        o = orc.OraclePlanner(scene, mgr_init_pos=scene["start"])
        obs, dt, cg, ws = scene["obstacles"], scene["dt"], scene["cost_gains"], scene["ws_limits"]
        flops = 0
        L.orc_flops_reset()
        for t in range(ticks):
            if episode and t % episode == 0:
                o.set_initial_position(scene["start"])
            best = o.evaluate(cg, ws)
            o.move_real(obs, dt, 1, best)
            pos, vel, _ = o.real_state()
            o.reset_agents(pos, vel, obs)
            f0 = L.orc_flops()
            o.rollout()
            flops += L.orc_flops() - f0
        steps, in_shell = L.orc_steps_total(), L.orc_steps_in_shell()
        paths, n = o.paths()
        o.close()
    finally:
        orc._LIB = saved
    return {"flops_per_agent_step": flops / max(steps, 1), "in_shell_step_fraction": in_shell / max(steps, 1),
            "agent_steps": int(steps), "ticks": ticks, "flops_total": int(flops),
            "convention": "+ - * / sqrt exp compare = 1 each; rollouts only (cfPrediction loop), reference "
                          "algorithm as restated in oracle/pmaf_oracle.c incl. its redundant evaluations",
            "final_paths_checksum": float(np.sum(paths)), "final_n_points": int(np.sum(n))}
