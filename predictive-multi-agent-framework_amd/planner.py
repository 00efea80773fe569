"""ctypes host binding of libpmaf_hip.so (include/pmaf.h).

Python is only the test/bench orchestration layer of this build; the product
host layer is the C++ facade in include/bimanual_planning_ros/. The binding is
deliberately thin: one method per C-ABI entry point, numpy in / numpy out,
every non-zero status raised as PmafError. There is no fallback: if the
shared library is missing or no HIP device is present this module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PMAF_LIB_PATH: load an alternative build of the same library (debug / timer builds)
# PMAF_VARIANT=rassoc: the library built with the other evaluation-order policy (csrc/build.sh, include/pmaf.h
# pmaf_eval_order) out of lib_rassoc/ -- the oracle binding (oracle/orc.py) follows the same variable
VARIANT = os.environ.get("PMAF_VARIANT", "")
LIB_PATH = os.environ.get("PMAF_LIB_PATH") or os.path.join(_HERE, "lib_" + VARIANT if VARIANT else "lib", "libpmaf_hip.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class PmafError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("pmaf status %d: %s" % (code, msg))
        self.code = code


class PmafParams(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("n_populations", C.c_int32), ("n_agents", C.c_int32),
        ("n_obstacles", C.c_int32), ("max_prediction_steps", C.c_int32), ("device", C.c_int32),
        ("lanes_per_agent", C.c_int32), ("flags", C.c_int32),
        ("dt", C.c_double), ("velocity_max", C.c_double), ("approach_dist", C.c_double),
        ("detect_shell_rad", C.c_double), ("agent_mass", C.c_double), ("radius", C.c_double),
        ("goal", _dp), ("init_pos", _dp), ("obstacles", _dp),
        ("k_attr", _dp), ("k_circ", _dp), ("k_repel", _dp), ("k_damp", _dp),
        ("agent_types", _ip), ("random_vecs", _dp),
    ]


# every symbol include/pmaf.h declares: name -> (restype, argtypes)
_V = C.c_void_p
SYMBOLS = {
    "pmaf_create": (C.c_int, [C.POINTER(PmafParams), C.POINTER(_V)]),
    "pmaf_destroy": (C.c_int, [_V]),
    "pmaf_last_error": (C.c_char_p, []),
    "pmaf_abi_version": (C.c_int, []),
    "pmaf_eval_order": (C.c_int, []),
    "pmaf_set_initial_position": (C.c_int, [_V, _dp]),
    "pmaf_set_real_position": (C.c_int, [_V, _dp]),
    "pmaf_start": (C.c_int, [_V]),
    "pmaf_stop": (C.c_int, [_V]),
    "pmaf_evaluate": (C.c_int, [_V, _dp, _dp, _ip]),
    "pmaf_move_real": (C.c_int, [_V, _dp, C.c_double, C.c_int32, _ip]),
    "pmaf_reset_agents": (C.c_int, [_V, _dp, _dp, _dp]),
    "pmaf_tick": (C.c_int, [_V, _dp, C.c_double, _dp, _dp, _ip, _dp, _dp]),
    "pmaf_link_force": (C.c_int, [_V, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp]),
    "pmaf_move_agents": (C.c_int, [_V, _dp, C.c_double, C.c_int32]),
    "pmaf_move_agent": (C.c_int, [_V, _dp, C.c_double, C.c_int32, _ip, C.c_int32, _ip]),
    "pmaf_set_agent_positions": (C.c_int, [_V, _dp]),
    "pmaf_set_agent_pos_and_vels": (C.c_int, [_V, _dp, _dp]),
    "pmaf_eval_obstacle_distance": (C.c_int, [_V, _dp, _dp]),
    "pmaf_get_paths": (C.c_int, [_V, _dp, _ip]),
    "pmaf_view_paths": (C.c_int, [_V, C.POINTER(_dp), C.POINTER(_ip)]),
    "pmaf_get_costs": (C.c_int, [_V, _dp]),
    "pmaf_get_path_lengths": (C.c_int, [_V, _dp]),
    "pmaf_get_min_obs_dist": (C.c_int, [_V, _dp]),
    "pmaf_get_success": (C.c_int, [_V, _ip]),
    "pmaf_get_agent_velocities": (C.c_int, [_V, _dp]),
    "pmaf_get_rotation_vectors": (C.c_int, [_V, _dp, _ip]),
    "pmaf_get_real_state": (C.c_int, [_V, _dp, _dp, _dp]),
    "pmaf_get_real_known": (C.c_int, [_V, _ip, _dp]),
    "pmaf_get_real_path": (C.c_int, [_V, C.c_int32, _dp, C.c_int32, _ip]),
    "pmaf_get_dist_from_goal": (C.c_int, [_V, _dp]),
    "pmaf_get_best": (C.c_int, [_V, _ip, _ip]),
    "pmaf_get_prediction_times_ns": (C.c_int, [_V, _dp]),
    "pmaf_set_best": (C.c_int, [_V, _ip, _ip, _dp]),
    "pmaf_write_winner_records": (C.c_int, [_V, _V, C.c_size_t]),
    "pmaf_winner_record_doubles": (C.c_size_t, [_V]),
    "pmaf_stream": (_V, [_V]),
    "pmaf_comm_unique_id": (C.c_int, [_V]),
    "pmaf_comm_init_rccl": (C.c_int, [C.c_int32, C.c_int32, _V, C.c_int32, C.POINTER(_V)]),
    "pmaf_comm_from_rccl": (C.c_int, [_V, C.c_int32, C.POINTER(_V)]),
    "pmaf_comm_init_host": (C.c_int, [C.c_int32, C.c_int32, _V, _V, C.POINTER(_V)]),
    "pmaf_comm_destroy": (C.c_int, [_V]),
    "pmaf_comm_world": (C.c_int, [_V]),
    "pmaf_comm_rank": (C.c_int, [_V]),
    "pmaf_comm_allgather": (C.c_int, [_V, _dp, _dp, C.c_size_t]),
    "pmaf_select_best": (C.c_int32, [_dp, C.c_int32, C.c_int32]),
    "pmaf_allgather_winners": (C.c_int, [_V, _V, _V, C.c_size_t]),
    "pmaf_attach_comm": (C.c_int, [_V, _V]),
    "pmaf_winners_wait": (C.c_int, [_V, C.POINTER(_dp), C.POINTER(C.c_size_t)]),
    "pmaf_winners_device": (_V, [_V]),
    "pmaf_get_exchange_times_us": (C.c_int, [_V, _dp, C.c_int32, _ip]),
    "pmaf_get_tick_times_us": (C.c_int, [_V, _dp, _dp, C.c_int32, _ip]),
    "pmaf_peer_export": (C.c_int, [_V, C.c_int32, _V]),
    "pmaf_peer_connect": (C.c_int, [_V, C.c_int32, C.c_int32, _V]),
    "pmaf_peer_couple": (C.c_int, [_V, C.c_int32, C.c_int32, C.c_int32, C.c_double, _dp]),
    "pmaf_peer_disconnect": (C.c_int, [_V]),
    "pmaf_peer_read": (C.c_int, [_V, _dp, _dp]),
    "pmaf_get_peer_times_us": (C.c_int, [_V, _dp, _dp, C.c_int32, _ip]),
    "pmaf_device_count": (C.c_int, []),
    "pmaf_state_size": (C.c_size_t, [_V]),
    "pmaf_save_state": (C.c_int, [_V, _V, C.c_size_t]),
    "pmaf_load_state": (C.c_int, [_V, _V, C.c_size_t]),
    "pmaf_set_profiling": (C.c_int, [_V, C.c_int32]),
    "pmaf_get_kernel_stats": (C.c_int, [_V, _dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pmaf_reset_kernel_stats": (C.c_int, [_V]),
    "pmaf_get_launch_count": (C.c_int, [_V, C.POINTER(C.c_int64)]),
    "pmaf_get_launch_config": (C.c_int, [_V, _ip, _ip, _ip]),
    "pmaf_get_waves_per_agent": (C.c_int, [_V, _ip, _ip]),
    "pmaf_get_priority_slices": (C.c_int, [_V, _ip, _ip, _ip]),
    "pmaf_pick_lanes_per_agent": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "pmaf_estimate_rollout_us": (C.c_double, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "pmaf_debug_math": (C.c_int, [C.c_int32, C.c_int32, _dp, _dp, _dp]),
    "pmaf_debug_external_rollout": (C.c_int, [_V, C.c_char_p, C.c_char_p]),
    "pmaf_get_health": (C.c_int, [_V, _ip]),
    "pmaf_enable_winner_path": (C.c_int, [_V, C.c_int32]),
    "pmaf_view_winner_path": (C.c_int, [_V, C.POINTER(_dp), C.POINTER(_ip), C.POINTER(_ip)]),
    "pmaf_get_winner_path_times_us": (C.c_int, [_V, _dp, C.c_int32, _ip]),
    "pmaf_debug_withhold_mailbox": (C.c_int, [_V, C.c_int32]),
    "pmaf_peer_info": (C.c_int, [_V, _ip, _ip]),
}


def debug_math(op, a, b=None):
    """evaluate an elementary op on the GPU (see pmaf_debug_math)"""
    L = load_library()
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(a if b is None else b, dtype=np.float64)
    out = np.zeros_like(a)
    rc = L.pmaf_debug_math(op, a.size, a.ctypes.data_as(_dp), b.ctypes.data_as(_dp), out.ctypes.data_as(_dp))
    if rc != 0:
        raise PmafError(rc, L.pmaf_last_error().decode())
    return out

_LIB = None


def load_library(path=None):
    """dlopen libpmaf_hip.so and type every exported symbol. Raises OSError if
    the library has not been built (python __graft_entry__.py build)."""
    global _LIB
    if _LIB is None:
        path = path or LIB_PATH
        if not os.path.exists(path):
            raise OSError("libpmaf_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(expected at %s); there is no CPU fallback" % path)
        lib = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def _d(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = np.ascontiguousarray(np.broadcast_to(a, shape))
    return a


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _pi(a):
    return a.ctypes.data_as(_ip) if a is not None else None


class PmafPlanner:
    """P populations of N agents on one MI355X. `scenes` is one scene dict or
    a list of P scene dicts (see scenes.py) sharing N, n_obs, capacity and the
    scalar parameters. Method names follow the C-ABI / CfManager."""

    def __init__(self, scenes, device=-1, lanes_per_agent=0, mgr_init_pos=None, fast_math=False,
                 ieee_sequences=False, blocking_wait=False, contracted=False):
        if isinstance(scenes, dict):
            scenes = [scenes]
        self.L = load_library()
        s0 = scenes[0]
        self.P = len(scenes)
        self.N = int(s0["n_agents"])
        self.n_obs = int(s0["obstacles"].shape[0])
        self.cap = int(s0["max_prediction_steps"])
        P, N, n_obs = self.P, self.N, self.n_obs
        for s in scenes:
            assert int(s["n_agents"]) == N and s["obstacles"].shape[0] == n_obs and int(s["max_prediction_steps"]) == self.cap
        prm = PmafParams()
        prm.abi_version = self.L.pmaf_abi_version()
        prm.n_populations, prm.n_agents, prm.n_obstacles = P, N, n_obs
        prm.max_prediction_steps = self.cap
        prm.device = device
        prm.lanes_per_agent = lanes_per_agent
        # PMAF_FLAG_FAST_MATH / _IEEE_SEQUENCES / _BLOCKING_WAIT / _CONTRACTED
        prm.flags = ((1 if fast_math else 0) | (2 if ieee_sequences else 0) | (4 if blocking_wait else 0)
                     | (8 if contracted else 0))
        prm.dt = s0["dt"]
        prm.velocity_max = s0["velocity_max"]
        prm.approach_dist = s0["approach_dist"]
        prm.detect_shell_rad = s0["detect_shell_rad"]
        prm.agent_mass = s0.get("agent_mass", 1.0)
        prm.radius = s0.get("radius", 0.05)
        keep = []
        goal = _d(np.stack([s["goal"] for s in scenes])); keep.append(goal)
        prm.goal = _p(goal)
        if mgr_init_pos is not None:
            ip = _d(mgr_init_pos, (P, 3)); keep.append(ip)
            prm.init_pos = _p(ip)
        obs = _d(np.stack([s["obstacles"] for s in scenes])); keep.append(obs)
        prm.obstacles = _p(obs)
        for k in ("k_attr", "k_circ", "k_repel", "k_damp"):
            g = _d(np.stack([np.broadcast_to(np.asarray(s[k], dtype=np.float64), (N,)) for s in scenes]))
            keep.append(g)
            setattr(prm, k, _p(g))
        types = s0.get("agent_types")
        if types is not None:
            types = np.ascontiguousarray(types, dtype=np.int32); keep.append(types)
            prm.agent_types = _pi(types)
        rv = _d(np.stack([s["random_vecs"] for s in scenes])); keep.append(rv)
        assert rv.shape == (P, N, n_obs, 3)
        prm.random_vecs = _p(rv)
        h = _V()
        self._h = None
        self._chk(self.L.pmaf_create(C.byref(prm), C.byref(h)))
        self._h = h

    # -- plumbing --
    def _chk(self, rc):
        if rc != 0:
            raise PmafError(rc, self.L.pmaf_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            self.L.pmaf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _sq(self, a):
        """drop the population axis for single-population handles"""
        return a[0] if self.P == 1 else a

    # -- CfManager surface --
    def set_initial_position(self, pos):
        a = _d(pos, (self.P, 3))
        self._chk(self.L.pmaf_set_initial_position(self._h, _p(a)))

    def set_real_position(self, pos):
        a = _d(pos, (self.P, 3))
        self._chk(self.L.pmaf_set_real_position(self._h, _p(a)))

    def start(self):
        self._chk(self.L.pmaf_start(self._h))

    def stop(self):
        self._chk(self.L.pmaf_stop(self._h))

    def rollout(self):
        """start + stop: run every agent's prediction to its guard."""
        self.start()
        self.stop()

    def evaluate(self, cost_gains, ws):
        g, w = _d(cost_gains), _d(ws)
        best = np.zeros(self.P, dtype=np.int32)
        self._chk(self.L.pmaf_evaluate(self._h, _p(g), _p(w), _pi(best)))
        return int(best[0]) if self.P == 1 else best

    def _obs(self, obstacles):
        if obstacles is None:
            return None
        return _d(obstacles).reshape(self.P, self.n_obs, 7)

    def move_real(self, obstacles, dt, steps, agent_id):
        o = self._obs(obstacles)
        ids = np.ascontiguousarray(np.broadcast_to(np.asarray(agent_id, dtype=np.int32), (self.P,)))
        self._chk(self.L.pmaf_move_real(self._h, _p(o), float(dt), int(steps), _pi(ids)))

    def reset_agents(self, pos, vel, obstacles):
        p, v, o = _d(pos, (self.P, 3)), _d(vel, (self.P, 3)), self._obs(obstacles)
        self._chk(self.L.pmaf_reset_agents(self._h, _p(p), _p(v), _p(o)))

    def tick(self, obstacles, dt, cost_gains, ws, want_outputs=True):
        o = self._obs(obstacles)
        g, w = _d(cost_gains), _d(ws)
        best = np.zeros(self.P, dtype=np.int32)
        npos = np.zeros((self.P, 3))
        nvel = np.zeros((self.P, 3))
        self._chk(self.L.pmaf_tick(self._h, _p(o), float(dt), _p(g), _p(w), _pi(best), _p(npos), _p(nvel)))
        self.last_next_pos, self.last_next_vel = npos, nvel
        return int(best[0]) if self.P == 1 else best

    # -- synchronous stepping API (a18) --
    def move_agents(self, obstacles, dt, steps):
        self._chk(self.L.pmaf_move_agents(self._h, _p(self._obs(obstacles)), float(dt), int(steps)))

    def move_agent(self, obstacles, dt, steps, agent_id, max_calls=1 << 30):
        ids = np.ascontiguousarray(np.broadcast_to(np.asarray(agent_id, dtype=np.int32), (self.P,)))
        calls = np.zeros(self.P, dtype=np.int32)
        self._chk(self.L.pmaf_move_agent(self._h, _p(self._obs(obstacles)), float(dt), int(steps), _pi(ids),
                                         int(min(max_calls, 2**31 - 1)), _pi(calls)))
        return int(calls[0]) if self.P == 1 else calls

    def set_agent_positions(self, pos):
        self._chk(self.L.pmaf_set_agent_positions(self._h, _p(_d(pos, (self.P, 3)))))

    def set_agent_pos_and_vels(self, pos, vel):
        self._chk(self.L.pmaf_set_agent_pos_and_vels(self._h, _p(_d(pos, (self.P, 3))), _p(_d(vel, (self.P, 3)))))

    def eval_obstacle_distance(self, obstacles):
        out = np.zeros((self.P, self.N))
        self._chk(self.L.pmaf_eval_obstacle_distance(self._h, _p(self._obs(obstacles)), _p(out)))
        return self._sq(out)

    def link_force(self, link_pos, k_r_force, obstacles, pop=0):
        lp, k, o = _d(link_pos), _d(k_r_force), self._obs(obstacles)
        out = np.zeros_like(lp)
        self._chk(self.L.pmaf_link_force(self._h, pop, lp.shape[0], _p(lp), _p(k), _p(o), _p(out)))
        return out

    # -- getters --
    def paths(self):
        paths = np.zeros((self.P, self.N, self.cap, 3))
        n = np.zeros((self.P, self.N), dtype=np.int32)
        self._chk(self.L.pmaf_get_paths(self._h, _p(paths), _pi(n)))
        return self._sq(paths), self._sq(n)

    def n_points(self):
        n = np.zeros((self.P, self.N), dtype=np.int32)
        self._chk(self.L.pmaf_get_paths(self._h, None, _pi(n)))
        return self._sq(n)

    def _get(self, fn, shape, dtype=np.float64):
        out = np.zeros(shape, dtype=dtype)
        ptr = _p(out) if dtype == np.float64 else _pi(out)
        self._chk(getattr(self.L, fn)(self._h, ptr))
        return self._sq(out)

    def costs(self):
        return self._get("pmaf_get_costs", (self.P, self.N))

    def path_lengths(self):
        return self._get("pmaf_get_path_lengths", (self.P, self.N))

    def min_obs_dist(self):
        return self._get("pmaf_get_min_obs_dist", (self.P, self.N))

    def success(self):
        return self._get("pmaf_get_success", (self.P, self.N), np.int32)

    def agent_vel(self):
        return self._get("pmaf_get_agent_velocities", (self.P, self.N, 3))

    def rot_vecs(self):
        rot = np.zeros((self.P, self.N, self.n_obs, 3))
        self._chk(self.L.pmaf_get_rotation_vectors(self._h, _p(rot), None))
        return self._sq(rot)

    def known(self):
        known = np.zeros((self.P, self.N, self.n_obs), dtype=np.int32)
        self._chk(self.L.pmaf_get_rotation_vectors(self._h, None, _pi(known)))
        return self._sq(known)

    def real_state(self):
        pos, vel, force = np.zeros((self.P, 3)), np.zeros((self.P, 3)), np.zeros((self.P, 3))
        self._chk(self.L.pmaf_get_real_state(self._h, _p(pos), _p(vel), _p(force)))
        return self._sq(pos), self._sq(vel), self._sq(force)

    def real_known(self):
        known = np.zeros((self.P, self.n_obs), dtype=np.int32)
        rot = np.zeros((self.P, self.n_obs, 3))
        self._chk(self.L.pmaf_get_real_known(self._h, _pi(known), _p(rot)))
        return self._sq(known), self._sq(rot)

    def real_path(self, pop=0):
        n = C.c_int32(0)
        self._chk(self.L.pmaf_get_real_path(self._h, pop, None, 0, C.byref(n)))
        out = np.zeros((n.value, 3))
        self._chk(self.L.pmaf_get_real_path(self._h, pop, _p(out), n.value, None))
        return out

    def dist_from_goal(self):
        out = self._get("pmaf_get_dist_from_goal", (self.P,))
        return float(out) if self.P == 1 else out

    def best(self):
        t = np.zeros(self.P, dtype=np.int32)
        i = np.zeros(self.P, dtype=np.int32)
        self._chk(self.L.pmaf_get_best(self._h, _pi(t), _pi(i)))
        return t, i

    def best_type(self):
        return int(self.best()[0][0])

    def best_id(self):
        return int(self.best()[1][0])

    def set_best(self, ids, types, rand_vecs=None):
        ids = np.ascontiguousarray(np.broadcast_to(np.asarray(ids, dtype=np.int32), (self.P,)))
        types = np.ascontiguousarray(np.broadcast_to(np.asarray(types, dtype=np.int32), (self.P,)))
        rv = _d(rand_vecs).reshape(self.P, self.n_obs, 3) if rand_vecs is not None else None
        self._chk(self.L.pmaf_set_best(self._h, _pi(ids), _pi(types), _p(rv)))

    def prediction_times_ns(self):
        return self._get("pmaf_get_prediction_times_ns", (self.P, self.N))

    # -- sharding / measurement --
    def winner_record_doubles(self):
        return int(self.L.pmaf_winner_record_doubles(self._h))

    def write_winner_records(self, device_ptr, nbytes):
        self._chk(self.L.pmaf_write_winner_records(self._h, C.c_void_p(device_ptr), nbytes))

    def allgather_winners(self, comm, recv_device_ptr, nbytes):
        """one-shot: pack + all-gather into device memory, stream-ordered on the handle's stream (RCCL)"""
        self._chk(self.L.pmaf_allgather_winners(self._h, comm._c, C.c_void_p(recv_device_ptr), nbytes))

    def attach_comm(self, comm):
        """every tick / evaluate from now on all-gathers its winner records (comm=None detaches)"""
        self._chk(self.L.pmaf_attach_comm(self._h, comm._c if comm is not None else None))
        self._comm = comm  # keep it alive while attached

    def winners_wait(self):
        """[world][P][record] table of the last tick's / evaluate's exchange (a copy)"""
        ptr = _dp()
        n = C.c_size_t(0)
        self._chk(self.L.pmaf_winners_wait(self._h, C.byref(ptr), C.byref(n)))
        a = np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()
        rec = self.winner_record_doubles()
        return a.reshape(-1, self.P, rec)

    def winners_device_ptr(self):
        return self.L.pmaf_winners_device(self._h)

    def exchange_times_us(self, max_n=1 << 20):
        out = np.zeros(max_n)
        n = C.c_int32(0)
        self._chk(self.L.pmaf_get_exchange_times_us(self._h, _p(out), max_n, C.byref(n)))
        return out[:n.value].copy()

    def tick_times_us(self, max_n=8192):
        """(enqueue_us, setpoint_us) of the newest pmaf_tick calls, measured inside the library; clears the record"""
        enq, sp = np.zeros(max_n), np.zeros(max_n)
        n = C.c_int32(0)
        self._chk(self.L.pmaf_get_tick_times_us(self._h, _p(enq), _p(sp), max_n, C.byref(n)))
        return enq[:n.value].copy(), sp[:n.value].copy()

    # -- failure detection / the selected trajectory on the host (ABI 5) --
    HEALTH_SETPOINT_NAN, HEALTH_FORCE_NAN, HEALTH_ACC_CLAMPED, HEALTH_COST_NAN = 1, 2, 4, 8

    def health(self):
        """PMAF_HEALTH_* bits of the last tick / evaluate / move_real, per population"""
        b = np.zeros(self.P, dtype=np.int32)
        self._chk(self.L.pmaf_get_health(self._h, _pi(b)))
        return self._sq(b)

    def enable_winner_path(self, enable=True):
        self._chk(self.L.pmaf_enable_winner_path(self._h, 1 if enable else 0))

    def winner_path(self):
        """(paths, n_points, agent) of the last selection: a list of P arrays [n_points[p]][3] copied out of the pinned
        buffer the manager kernel wrote (pmaf_view_winner_path)"""
        pp, pn, pa = _dp(), _ip(), _ip()
        self._chk(self.L.pmaf_view_winner_path(self._h, C.byref(pp), C.byref(pn), C.byref(pa)))
        n = np.ctypeslib.as_array(pn, shape=(self.P,)).copy()
        a = np.ctypeslib.as_array(pa, shape=(self.P,)).copy()
        full = np.ctypeslib.as_array(pp, shape=(self.P, self.cap, 3))
        paths = [full[p, :n[p]].copy() for p in range(self.P)]
        return (paths[0], int(n[0]), int(a[0])) if self.P == 1 else (paths, n, a)

    def winner_path_wait(self):
        """pmaf_view_winner_path without the copies (latency measurements)"""
        self._chk(self.L.pmaf_view_winner_path(self._h, None, None, None))

    def winner_path_times_us(self, max_n=1 << 16):
        out = np.zeros(max_n)
        n = C.c_int32(0)
        self._chk(self.L.pmaf_get_winner_path_times_us(self._h, _p(out), max_n, C.byref(n)))
        return out[:n.value].copy()

    def debug_withhold_mailbox(self, enable=True):
        self._chk(self.L.pmaf_debug_withhold_mailbox(self._h, 1 if enable else 0))

    def peer_info(self):
        f, w = C.c_int32(0), C.c_int32(0)
        self._chk(self.L.pmaf_peer_info(self._h, C.byref(f), C.byref(w)))
        return {"fine_grained": bool(f.value), "world": w.value}

    # -- peer mailboxes (header-only exchange without a collective) --
    PEER_HANDLE_BYTES = 128

    def peer_export(self, world):
        buf = (C.c_ubyte * self.PEER_HANDLE_BYTES)()
        self._chk(self.L.pmaf_peer_export(self._h, int(world), C.cast(buf, C.c_void_p)))
        return bytes(buf)

    def peer_connect(self, world, rank, handles):
        """handles: list of `world` byte strings from peer_export, in rank order"""
        blob = b"".join(handles)
        assert len(blob) == world * self.PEER_HANDLE_BYTES
        buf = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
        self._chk(self.L.pmaf_peer_connect(self._h, int(world), int(rank), C.cast(buf, C.c_void_p)))
        self._peer_world = world

    def peer_couple(self, pop, src_rank, src_pop, radius, init_pos=None):
        ip = _d(init_pos, (3,)) if init_pos is not None else None
        self._chk(self.L.pmaf_peer_couple(self._h, int(pop), int(src_rank), int(src_pop), float(radius), _p(ip)))

    def peer_disconnect(self):
        self._chk(self.L.pmaf_peer_disconnect(self._h))

    def peer_read(self):
        """(headers [world][P][8], seq [world][P]) -- the newest header of every population in this rank's inbox"""
        w = self._peer_world
        hd, sq = np.zeros((w, self.P, 8)), np.zeros((w, self.P))
        self._chk(self.L.pmaf_peer_read(self._h, _p(hd), _p(sq)))
        return hd, sq

    def peer_times_us(self, max_n=1 << 20):
        w, pb = np.zeros(max_n), np.zeros(max_n)
        n = C.c_int32(0)
        self._chk(self.L.pmaf_get_peer_times_us(self._h, _p(w), _p(pb), max_n, C.byref(n)))
        return w[:n.value].copy(), pb[:n.value].copy()

    def save_state(self):
        n = int(self.L.pmaf_state_size(self._h))
        buf = np.zeros(n, dtype=np.uint8)
        self._chk(self.L.pmaf_save_state(self._h, buf.ctypes.data_as(C.c_void_p), n))
        return buf

    def load_state(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self._chk(self.L.pmaf_load_state(self._h, blob.ctypes.data_as(C.c_void_p), blob.size))

    def stream(self):
        return self.L.pmaf_stream(self._h)

    def set_profiling(self, on=True):
        """on: False / True, or n > 1 = time every n-th rollout launch"""
        self._chk(self.L.pmaf_set_profiling(self._h, int(on) if (on is not True and on is not False and int(on) > 1) else (1 if on else 0)))

    def kernel_stats(self):
        ms = C.c_double(0)
        n = C.c_int64(0)
        steps = C.c_int64(0)
        self._chk(self.L.pmaf_get_kernel_stats(self._h, C.byref(ms), C.byref(n), C.byref(steps)))
        return ms.value, n.value, steps.value

    def launch_count(self):
        n = C.c_int64(0)
        self._chk(self.L.pmaf_get_launch_count(self._h, C.byref(n)))
        return n.value

    def reset_kernel_stats(self):
        self._chk(self.L.pmaf_reset_kernel_stats(self._h))

    def external_rollout(self, code_object_path, kernel_name=None):
        """measurement tooling: run the rollout launches out of an external code object (None: built-in again)"""
        self._chk(self.L.pmaf_debug_external_rollout(
            self._h, code_object_path.encode() if code_object_path else None, kernel_name.encode() if kernel_name else None))

    def launch_config(self):
        a, b, c = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._chk(self.L.pmaf_get_launch_config(self._h, C.byref(a), C.byref(b), C.byref(c)))
        w, pw = C.c_int32(0), C.c_int32(0)
        self._chk(self.L.pmaf_get_waves_per_agent(self._h, C.byref(w), C.byref(pw)))
        ps = C.c_int32(0)
        self._chk(self.L.pmaf_get_priority_slices(self._h, C.byref(ps), None, None))
        return dict(lanes_per_agent=a.value, n_blocks=b.value, lds_bytes=c.value, waves_per_agent=w.value,
                    obstacles_per_wave=pw.value, priority_slices=bool(ps.value))


HOST_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


class PmafComm:
    """pmaf_comm: RCCL communicator (one rank per GPU) or a host-transport one
    whose all-gather is a Python callable (gloo / MPI in tests)."""

    def __init__(self):
        self.L = load_library()
        self._c = None
        self._cb = None

    @staticmethod
    def unique_id():
        L = load_library()
        buf = (C.c_ubyte * 128)()
        rc = L.pmaf_comm_unique_id(C.cast(buf, C.c_void_p))
        if rc != 0:
            raise PmafError(rc, L.pmaf_last_error().decode())
        return bytes(buf)

    @classmethod
    def rccl(cls, world, rank, unique_id, device=-1):
        self = cls()
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        c = _V()
        rc = self.L.pmaf_comm_init_rccl(world, rank, C.cast(buf, C.c_void_p), device, C.byref(c))
        if rc != 0:
            raise PmafError(rc, self.L.pmaf_last_error().decode())
        self._c = c
        return self

    @classmethod
    def from_rccl(cls, nccl_comm_ptr, device=-1):
        """adopt an ncclComm_t the application owns (pmaf_comm_from_rccl; not destroyed with this object)"""
        self = cls()
        c = _V()
        rc = self.L.pmaf_comm_from_rccl(C.c_void_p(nccl_comm_ptr), device, C.byref(c))
        if rc != 0:
            raise PmafError(rc, self.L.pmaf_last_error().decode())
        self._c = c
        return self

    @classmethod
    def host(cls, world, rank, allgather):
        """allgather(send: np.ndarray[uint8]) -> np.ndarray[uint8] of world * len(send) bytes, rank-major"""
        self = cls()

        def _cb(ctx, send, recv, nbytes):
            try:
                snd = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_ubyte)), shape=(nbytes,))
                out = np.ascontiguousarray(allgather(snd.copy()), dtype=np.uint8).reshape(-1)
                if out.size != nbytes * world:
                    return 2
                C.memmove(recv, out.ctypes.data, out.size)
                return 0
            except Exception:  # never unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1

        self._cb = HOST_ALLGATHER_FN(_cb)
        c = _V()
        rc = self.L.pmaf_comm_init_host(world, rank, C.cast(self._cb, C.c_void_p), None, C.byref(c))
        if rc != 0:
            raise PmafError(rc, self.L.pmaf_last_error().decode())
        self._c = c
        return self

    @property
    def world(self):
        return self.L.pmaf_comm_world(self._c)

    @property
    def rank(self):
        return self.L.pmaf_comm_rank(self._c)

    def allgather(self, a):
        """blocking all-gather of a host float64 array; returns [world, *a.shape]"""
        a = np.ascontiguousarray(a, dtype=np.float64)
        out = np.zeros((self.world,) + a.shape)
        rc = self.L.pmaf_comm_allgather(self._c, _p(a), _p(out), a.size)
        if rc != 0:
            raise PmafError(rc, self.L.pmaf_last_error().decode())
        return out

    def close(self):
        if self._c:
            self.L.pmaf_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count():
    """hipGetDeviceCount as the library sees it"""
    return int(load_library().pmaf_device_count())


def select_best(costs, prev_best=None):
    """pmaf_select_best: evaluateAgents' selection rule on a gathered cost vector"""
    L = load_library()
    c = np.ascontiguousarray(costs, dtype=np.float64)
    return int(L.pmaf_select_best(_p(c), c.size, -1 if prev_best is None else int(prev_best)))
