#!/usr/bin/env python3
"""Replays one reference-pin scenario (oracle/pin/scenarios/<name>.txt) on the CPU oracle -- the same call sequence
oracle/pin/pin_harness.cpp drives the reference's CfManager through -- and compares with the reference's record
tests/golden/ref_<name>.json bit for bit. The oracle build (dot-product association) follows PMAF_VARIANT like the rest
of the test infrastructure; std::exp is the platform libm on both sides (oracle mode 0).

The fixture's keys come from fixture_schema.py (the one definition both this file and the harness are held to). With
ref = None, replay() RECORDS instead of comparing: the same call sequence writes a fixture in the harness's schema from
whatever planner it drives. That exists to validate this tool chain (tests/test_reference_pin.py: record from the oracle
into a temporary directory, replay, corrupt single values, see exactly those reported) -- a fixture recorded from the
oracle pins nothing and is never written to tests/golden/ref_*.json, a name reserved for the reference's own output.

TEST INFRASTRUCTURE. usage: python oracle/pin/replay.py <scenario.txt> <ref.json>   -> one JSON line
  {"match": bool, "order": 0|1, "ticks": n, "compared": n_values, "mismatches": n, "first": {...}, "max_abs_diff": d,
   "fields": {what: {"count": n, "ticks": [first few ticks]}}}
       python oracle/pin/replay.py --record <scenario.txt> <out.json>          (tool validation only, see above)"""
import copy
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import fixture_schema as FS  # noqa: E402


def load_scenario(path):
    tok = open(path).read().split()
    pos = [0]

    def nxt():
        pos[0] += 1
        return tok[pos[0] - 1]

    def num():
        t = nxt()
        return float.fromhex(t) if "x" in t.lower() else float(t)

    assert nxt() == "pmaf-pin-scenario"
    num()
    s = {}
    while pos[0] < len(tok):
        k = nxt()
        if k == "name":
            s["name"] = nxt()
        elif k in ("n_agents", "n_body", "max_prediction_steps", "freq_multiple", "dynamic", "dump_paths", "detail_every"):
            s[k] = int(num())
        elif k in ("dt", "k_repel_body"):
            s[k] = num()
        elif k == "gains":
            s["k_attr"], s["k_circ"], s["k_repel"], s["k_damp"], s["k_manip"] = [num() for _ in range(5)]
        elif k == "limits":
            s["velocity_max"], s["approach_dist"], s["detect_shell_rad"] = [num() for _ in range(3)]
        elif k == "cost":
            s["cost_gains"] = np.array([num() for _ in range(4)])
        elif k == "ws":
            s["ws_limits"] = np.array([num() for _ in range(6)])
        elif k == "start":
            s["start"] = np.array([num() for _ in range(3)])
        elif k == "closed_loop":
            s["closed_loop"] = int(num())
            s["lag"] = num()
        elif k == "obstacles":
            n = int(num())
            s["obstacles"] = np.array([[num() for _ in range(7)] for _ in range(n)])
        elif k == "goals":
            n = int(num())
            s["goals"] = [(np.array([num() for _ in range(3)]), int(num()), int(num())) for _ in range(n)]
        elif k == "random":
            n = int(num())
            s["random"] = np.array([[num() for _ in range(3)] for _ in range(n)]).reshape(n, 3)
        else:
            raise ValueError("unknown scenario key " + k)
    return s


def normalise(raw, order):
    """Eigen's normalize(): z = squaredNorm() in the association under test; v /= sqrt(z) if z > 0"""
    x, y, z = raw[..., 0], raw[..., 1], raw[..., 2]
    zz = x * x + (y * y + z * z) if order == 1 else (x * x + y * y) + z * z
    s = np.sqrt(np.where(zz > 0, zz, 1.0))
    return raw / s[..., None]


def replay(scn, ref, make=None, tol=0.0, selected_only=False, order=None):
    """ref = a loaded fixture: compare; ref = None: RECORD (returns the fixture as a dict, key "fixture" of the result).
    make(scene, mgr_init_pos) -> planner with the OraclePlanner / PmafPlanner method surface (default: the CPU oracle in
    its libm-exp mode, compared bit for bit). tests/test_reference_pin.py also runs the HIP planner through it with
    tol = 1e-5 m and selected_only = True (its exp is portable_exp, not the reference's libm: the north star's contract
    is the selected trajectory / set-point sequence and the best-index sequence)."""
    if make is None:
        from oracle import orc
        orc.build()
        orc.set_exp_mode(0)
        order = orc.eval_order()
        make = lambda sc, ip: orc.OraclePlanner(sc, mgr_init_pos=ip)
    N, n_obs = scn["n_agents"], scn["obstacles"].shape[0]
    base = dict(n_agents=N, max_prediction_steps=scn["max_prediction_steps"], dt=scn["freq_multiple"] * scn["dt"],
                velocity_max=scn["velocity_max"], approach_dist=scn["approach_dist"], detect_shell_rad=scn["detect_shell_rad"],
                agent_mass=1.0, radius=0.05, k_attr=scn["k_attr"], k_circ=scn["k_circ"], k_repel=scn["k_repel"],
                k_damp=scn["k_damp"], cost_gains=scn["cost_gains"], ws_limits=scn["ws_limits"])
    stats = dict(compared=0, mismatches=0, first=None, max_abs_diff=0.0, ticks=0, fields={})
    recording = ref is None
    hx3 = lambda v: [float(x).hex() for x in np.asarray(v, dtype=np.float64).ravel()]
    out = {"format": FS.FORMAT, "scenario": scn["name"],
           "meta": dict({k: None for k in FS.META_KEYS}, compiler="oracle/pin/replay.py --record (NOT the reference)"), "goals": []}

    def note(what, tick, n_bad):
        f = stats["fields"].setdefault(what, {"count": 0, "ticks": []})
        f["count"] += int(n_bad)
        if len(f["ticks"]) < 8 and tick not in f["ticks"]:
            f["ticks"].append(tick)

    def cmp(what, tick, got, want_hex):
        got = np.asarray(got, dtype=np.float64).ravel()
        want = np.array([float.fromhex(x) for x in np.asarray(want_hex).ravel()])
        assert got.shape == want.shape, (what, tick, got.shape, want.shape)
        with np.errstate(invalid="ignore"):
            bad = ~((got == want) | (np.isnan(got) & np.isnan(want)) | (np.abs(got - want) <= tol))
        stats["compared"] += int(got.size)
        if bad.any():
            stats["mismatches"] += int(bad.sum())
            note(what, tick, bad.sum())
            with np.errstate(invalid="ignore"):
                d = np.nanmax(np.abs(got - want)[bad]) if np.isfinite((got - want)[bad]).any() else float("inf")
            stats["max_abs_diff"] = max(stats["max_abs_diff"], float(d))
            if stats["first"] is None:
                i = int(np.argmax(bad))
                stats["first"] = dict(what=what, tick=tick, index=i, oracle=float(got[i]).hex(), reference=float(want[i]).hex())

    def cmpi(what, tick, got, want):
        got, want = np.asarray(got).ravel(), np.asarray(want).ravel()
        stats["compared"] += int(got.size)
        if not np.array_equal(got, want):
            stats["mismatches"] += int((got != want).sum()) if got.shape == want.shape else 1
            note(what, tick, int((got != want).sum()) if got.shape == want.shape else 1)
            if stats["first"] is None:
                stats["first"] = dict(what=what, tick=tick, oracle=got.tolist()[:16], reference=want.tolist()[:16])

    obs = scn["obstacles"].copy()
    position = scn["start"].copy()
    ora, old_rv = None, None
    mgr_init = scn["start"].copy()
    for gi, (goal, max_ticks, until_reached) in enumerate(scn["goals"]):
        per_init = (N - 5) * n_obs if N > 5 else 0
        if recording:   # the reference's constructors draw (N - 5) * n_obs triples per init(), in order
            rg = {"goal": hx3(goal), "random_first": gi * per_init, "random_used": per_init, "ticks": []}
            out["goals"].append(rg)
        else:
            rg = ref["goals"][gi]
        assert rg["random_used"] == per_init, "the reference made another number of makeRandomVector() calls than expected"
        rv = np.zeros((N, n_obs, 3))
        if per_init:
            rv[5:] = normalise(scn["random"][rg["random_first"]:rg["random_first"] + per_init].reshape(N - 5, n_obs, 3), order)
        sc = dict(base, goal=goal, obstacles=obs.copy(), random_vecs=rv)
        if ora is None:
            new = make(sc, mgr_init)
            cur = scn["start"].copy()
        else:
            ora.set_initial_position(position)          # the position message while planning is inactive
            mgr_init = position.copy()
            cur = np.asarray(ora.real_state()[0]).copy()
            bid, btype = ora.best_id(), ora.best_type()
            new = make(sc, mgr_init)
            if bid > 0:
                new.set_best(bid, btype, old_rv[bid - 1])
            ora.close()
        ora, old_rv = new, rv
        ora.set_initial_position(cur)
        if recording:
            rg["start"] = hx3(cur)
        else:
            cmp("goal start", -1, cur, rg["start"])
        position = np.array([cur[0], cur[1], (cur[2] + 0.00001) - 0.00001])
        n_ticks = 0
        for t in range(max_ticks):
            if recording:
                rt = {}
                if scn.get("detail_every", 1) <= 1 or t % scn["detail_every"] == 0:
                    rt["n"] = None   # (filled below: this tick carries the per-agent detail)
            elif t >= len(rg["ticks"]):   # the reference's run ended earlier (e.g. it reached the goal)
                cmpi("run length of goal %d" % gi, t, [t + 1], [len(rg["ticks"])])
                break
            else:
                rt = rg["ticks"][t]
            if scn.get("closed_loop"):
                ora.set_real_position(position)
            scored = None
            if "n" in rt:   # what the selection of this tick scores
                if hasattr(ora, "stop"):
                    ora.stop()
                paths, n = ora.paths()
                scored = (paths, n)
                if recording:
                    rt["n"] = [int(x) for x in n]
                    rt["len"] = hx3(ora.path_lengths())
                    rt["reached"] = [int(bool(x)) for x in ora.success()]
                    rt["last"] = [hx3(paths[i, n[i] - 1]) for i in range(N)]
                    if scn.get("dump_paths"):
                        rt[FS.TICK_PATHS_KEY] = [[hx3(q) for q in paths[i, :n[i]]] for i in range(N)]
                elif not selected_only:
                    cmpi("n_steps", t, n, rt["n"])
                    cmp("path lengths", t, ora.path_lengths(), rt["len"])
                    cmpi("reached", t, ora.success(), rt["reached"])
                    cmp("last points", t, np.stack([paths[i, n[i] - 1] for i in range(N)]), rt["last"])
                    if "paths" in rt and np.array_equal(n, rt["n"]):
                        for i in range(N):
                            cmp("path of agent %d" % i, t, paths[i, :n[i]], rt["paths"][i])
            b = ora.tick(obs, scn["dt"], scn["cost_gains"], scn["ws_limits"])
            if recording:
                pos, vel, force = ora.real_state()
                rt.update(best=int(b), type=int(ora.best_type()), pos=hx3(pos), vel=hx3(vel), force=hx3(force),
                          dist=float(ora.dist_from_goal()).hex(), resumed=0)
                rg["ticks"].append(rt)
            elif selected_only and scored is not None and b == rt["best"]:   # the selected agent's scored trajectory
                paths, n = scored
                cmpi("n_steps of the selected agent", t, [n[b]], [rt["n"][b]])
                cmp("last point of the selected agent", t, paths[b, n[b] - 1], rt["last"][b])
                if "paths" in rt and n[b] == rt["n"][b]:
                    cmp("path of the selected agent", t, paths[b, :n[b]], rt["paths"][b])
            pos, vel, force = ora.real_state()
            if not recording:
                cmpi("best index", t, [b], [rt["best"]])
                cmpi("best type", t, [ora.best_type()], [rt["type"]])
                cmp("set-point", t, pos, rt["pos"])
                cmp("velocity", t, vel, rt["vel"])
                if not selected_only:
                    cmp("force", t, force, rt["force"])
                cmp("goal distance", t, [ora.dist_from_goal()], [rt["dist"]])
            nxt = np.asarray(pos).copy()
            position = nxt - scn["lag"] * (nxt - position) if scn.get("closed_loop") else nxt
            if scn["dynamic"]:
                obs = obs.copy()
                obs[:-1, 0:3] = obs[:-1, 0:3] + obs[:-1, 3:6] / 100.0
            n_ticks = t + 1
            if until_reached and ora.dist_from_goal() < 0.01:
                break
        if recording:
            rg["n_ticks"] = n_ticks
            rg["planned_trajectory"] = len(ora.real_path())
        else:
            cmpi("ticks of goal %d" % gi, -1, [n_ticks], [rg["n_ticks"]])
            cmpi("planned trajectory points", -1, [len(ora.real_path())], [rg["planned_trajectory"]])
        stats["ticks"] += n_ticks
    if hasattr(ora, "close"):
        ora.close()
    res = dict(stats, match=stats["mismatches"] == 0, order=order)
    if recording:
        res["fixture"] = out
    return res


if __name__ == "__main__":
    if sys.argv[1] == "--record":
        out_path = os.path.abspath(sys.argv[3])
        assert not os.path.basename(out_path).startswith("ref_") or os.path.dirname(out_path) != os.path.join(ROOT, "tests", "golden"), \
            "tests/golden/ref_*.json is reserved for what oracle/pin/make_pin.sh writes from the REFERENCE"
        res = replay(load_scenario(sys.argv[2]), None)
        json.dump(res.pop("fixture"), open(out_path, "w"))
        print(json.dumps(res))
        sys.exit(0)
    scn = load_scenario(sys.argv[1])
    ref = json.load(open(sys.argv[2]))
    assert ref["format"] == FS.FORMAT and ref["scenario"] == scn["name"]
    FS.check(ref)
    print(json.dumps(replay(scn, ref)))
