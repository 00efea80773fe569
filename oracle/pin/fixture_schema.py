"""The ONE definition of the reference-pin fixture format (tests/golden/ref_<scenario>.json).

oracle/pin/pin_harness.cpp WRITES it (the reference's CfManager driven through the node's call sequence), replay.py
READS it (and, in recording mode, writes it from any planner with the oracle's method surface -- tool validation only).
Both sides are held to this module: replay.py takes every key from here, and tests/test_reference_pin.py extracts the
keys of the harness's fprintf format strings and compares them with these sets, so a key renamed on one side cannot go
unnoticed until a maintainer's first real pin run. Every double is a C99 hex literal in a JSON string ("%a").

TEST INFRASTRUCTURE."""

FORMAT = "pmaf-reference-pin-1"

TOP_KEYS = ("format", "scenario", "meta", "goals")
# what the numbers depend on besides the sources (recorded, never compared)
META_KEYS = ("eigen", "eigen_vectorize", "eigen_dont_vectorize", "compiler", "optimize", "fma_contraction_possible", "glibc", "cpu_fma")
# per goal (one CfManager::init): goal / start are hex triples; random_first / random_used index the scenario's raw triples
GOAL_KEYS = ("goal", "start", "random_first", "random_used", "ticks", "n_ticks", "planned_trajectory")
# per tick, always: best index, best type, next set-point / velocity / force on the real agent (hex triples), goal distance
TICK_KEYS = ("best", "type", "pos", "vel", "force", "dist", "resumed")
# per tick when the scenario's detail_every selects it: what THIS tick's selection scored, per agent
TICK_DETAIL_KEYS = ("n", "len", "reached", "last")
# ... and with dump_paths: every agent's full path (hex triples)
TICK_PATHS_KEY = "paths"

ALL_KEYS = frozenset(TOP_KEYS + META_KEYS + GOAL_KEYS + TICK_KEYS + TICK_DETAIL_KEYS + (TICK_PATHS_KEY,))


def keys_written_by_harness(source_text):
    """the JSON keys that appear in pin_harness.cpp's format strings: every \\"name\\": inside a C string literal"""
    import re
    return frozenset(re.findall(r'\\"([a-z_]+)\\":', source_text))


def check(fixture):
    """structural check of a loaded fixture against this schema; returns the number of ticks"""
    assert tuple(sorted(fixture)) == tuple(sorted(TOP_KEYS)), sorted(fixture)
    assert fixture["format"] == FORMAT
    assert set(fixture["meta"]) == set(META_KEYS), sorted(fixture["meta"])
    ticks = 0
    for g in fixture["goals"]:
        assert set(g) == set(GOAL_KEYS), sorted(g)
        assert g["n_ticks"] == len(g["ticks"])
        for t in g["ticks"]:
            k = set(t)
            assert set(TICK_KEYS) <= k, sorted(k)
            extra = k - set(TICK_KEYS)
            assert extra in (set(), set(TICK_DETAIL_KEYS), set(TICK_DETAIL_KEYS) | {TICK_PATHS_KEY}), sorted(extra)
            for h in t["pos"] + t["vel"] + t["force"] + [t["dist"]]:
                float.fromhex(h)
        ticks += len(g["ticks"])
    return ticks
