"""Scene definitions for the planner tick: the reference's shipped task scenes
(values transcribed from its task YAML files, cited below) and the seeded
synthetic sphere scenes of SURVEY.md section 8(d) used by bench.py and the
parity tests.

A scene is a plain dict:
  n_agents, max_prediction_steps (path capacity in points = H+1), dt,
  start[3], goal[3], obstacles[n_obs][7] (px,py,pz,vx,vy,vz,r; LAST row is the
  repulsive-only obstacle, reference README.md:80), velocity_max,
  approach_dist, detect_shell_rad, agent_mass, radius, k_attr, k_circ, k_repel,
  k_damp, cost_gains[4] (k_goal_dist,k_path_len,k_safe_dist,k_workspace),
  ws_limits[6] ([xmax,xmin,ymax,ymin,zmax,zmin]), random_vecs[N][n_obs][3].
"""
import numpy as np

_MASK = (1 << 64) - 1
_GAMMA = 0x9E3779B97F4A7C15

# Random-agent type code etc. follow CfAgent::Type
# (B/include/bimanual_planning_ros/cf_agent.h:59-68)
REAL_AGENT, GOAL_HEURISTIC, OBSTACLE_HEURISTIC, GOAL_OBSTACLE_HEURISTIC, \
    VEL_HEURISTIC, RANDOM_AGENT, HAD_HEURISTIC = range(7)


def default_agent_types(n_agents):
    """Population layout of CfManager::init (B/src/cf_manager.cpp:70-104)."""
    head = [HAD_HEURISTIC, GOAL_HEURISTIC, OBSTACLE_HEURISTIC,
            GOAL_OBSTACLE_HEURISTIC, VEL_HEURISTIC]
    t = (head + [RANDOM_AGENT] * max(0, n_agents - 5))[:n_agents]
    return np.asarray(t, dtype=np.int32)


class SplitMix64:
    """Vectorised SplitMix64 -> doubles in [0,1): (u >> 11) * 2**-53."""

    def __init__(self, seed):
        self.state = int(seed) & _MASK

    def uniform(self, n):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        with np.errstate(over="ignore"):
            z = np.uint64(self.state) + idx * np.uint64(_GAMMA)
            self.state = int(z[-1]) if n else self.state
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def uniform_range(self, lo, hi, n):
        return lo + (hi - lo) * self.uniform(n)


def random_unit_vectors(rng, n_agents, n_obs):
    """Normalised U(-1,1)^3 vectors (RandomCfAgent ctor,
    B/include/bimanual_planning_ros/cf_agent.h:338-342) from an explicit
    generator instead of std::random_device."""
    v = rng.uniform_range(-1.0, 1.0, n_agents * n_obs * 3).reshape(n_agents, n_obs, 3)
    nrm = np.sqrt((v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2])
    nrm = np.where(nrm > 0, nrm, 1.0)
    return v / nrm[..., None]


# Gains/limits shared by the reference's dual-arm task files
# (B/config/tasks/dual_arms_static1.yaml:4-20).
_STATIC1_PARAMS = dict(
    dt=0.01, velocity_max=0.2, approach_dist=0.25, detect_shell_rad=0.35,
    agent_mass=1.0, radius=0.05,  # CfManager::init defaults, cf_manager.h:101-102
    k_attr=4.0, k_circ=0.025, k_repel=0.08, k_damp=3.0,
    cost_gains=np.array([100.0, 10.0, 0.001, 1.0]),
    ws_limits=np.array([1.0, -1.0, 0.3, -0.3, 1.1, 0.2]),
)

SENTINEL = [100.0, 100.0, 100.0, 0.0, 0.0, 0.0, 0.1]


def static1_obstacles():
    """The 9 spheres + repulsive sentinel of B/config/tasks/dual_arms_static1.yaml:38-69."""
    rows = []
    for x, z in ((0.125, 1.0), (0.125, 0.7), (-0.35, 0.6)):
        for y in (0.0, 0.125, -0.125):
            rows.append([x, y, z, 0.0, 0.0, 0.0, 0.1])
    rows.append(SENTINEL)
    return np.asarray(rows, dtype=np.float64)


def dyn1_obstacles():
    """B/config/tasks/dual_arms_dyn1.yaml:38-51 (3 moving spheres + sentinel)."""
    return np.asarray([
        [-0.2, 0.0, 0.9, 0.0, -0.0, 0.04, 0.2],
        [-0.2, 0.0, 0.3, 0.0, 0.0, 0.04, 0.225],
        [0.2, 0.0, -0.25, 0.0, 0.0, 0.1, 0.1],
        SENTINEL], dtype=np.float64)


def static1_scene(n_agents=16, horizon=100, seed=0xC0FFEE00 + 1 * 256, random_vecs=None):
    """BASELINE config C1: static1 scene, goal (0.5,0,0.7)
    (dual_arms_static1.yaml:85), start (-0.6,0,0.75) (build-chosen)."""
    obs = static1_obstacles()
    s = dict(_STATIC1_PARAMS)
    s.update(name="static1", n_agents=n_agents, max_prediction_steps=horizon + 1,
             start=np.array([-0.6, 0.0, 0.75]), goal=np.array([0.5, 0.0, 0.7]),
             obstacles=obs)
    if random_vecs is None:
        random_vecs = random_unit_vectors(SplitMix64(seed), n_agents, obs.shape[0])
    s["random_vecs"] = random_vecs
    return s


def dyn1_scene(n_agents=10, horizon=1500, seed=0xC0FFEE00 + 6 * 256, random_vecs=None):
    """dual_arms_dyn1.yaml: k_circ 0.015, k_damp 4 (:5,:7), moving obstacles."""
    obs = dyn1_obstacles()
    s = dict(_STATIC1_PARAMS)
    s.update(name="dyn1", n_agents=n_agents, max_prediction_steps=horizon + 1,
             k_circ=0.015, k_damp=4.0,
             start=np.array([-0.55, 0.0, 0.6]), goal=np.array([0.5, 0.0, 0.8]),
             obstacles=obs)
    if random_vecs is None:
        random_vecs = random_unit_vectors(SplitMix64(seed), n_agents, obs.shape[0])
    s["random_vecs"] = random_vecs
    return s


def synthetic_scene(n_agents, horizon, n_field_obstacles, config_id=2, scene_id=0,
                    dynamic=False, agent_types=None):
    """SURVEY.md 8(d) synthetic sphere scene. Deterministic in
    (config_id, scene_id). start (-0.6,0,0.7) -> goal (0.6,0,0.7); M spheres in
    x[-0.45,0.45] y[-0.3,0.3] z[0.4,1.0], r U[0.03,0.08], surface >= 0.10 m
    from start and goal; sentinel at (100,100,100) r 0.1."""
    rng = SplitMix64(0xC0FFEE00 + config_id * 256 + scene_id)
    start = np.array([-0.6, 0.0, 0.7])
    goal = np.array([0.6, 0.0, 0.7])
    rows = []
    while len(rows) < n_field_obstacles:
        u = rng.uniform(7)
        c = np.array([-0.45 + 0.9 * u[0], -0.3 + 0.6 * u[1], 0.4 + 0.6 * u[2]])
        r = 0.03 + 0.05 * u[3]
        if dynamic:
            v = -0.05 + 0.1 * u[4:7]
        else:
            v = np.zeros(3)
        if np.linalg.norm(c - start) - r < 0.10 or np.linalg.norm(c - goal) - r < 0.10:
            continue
        rows.append([c[0], c[1], c[2], v[0], v[1], v[2], r])
    rows.append(SENTINEL)
    obs = np.asarray(rows, dtype=np.float64)
    s = dict(_STATIC1_PARAMS)
    s.update(name="synthetic_c%d_s%d%s" % (config_id, scene_id, "_dyn" if dynamic else ""),
             n_agents=n_agents, max_prediction_steps=horizon + 1,
             start=start, goal=goal, obstacles=obs,
             random_vecs=random_unit_vectors(rng, n_agents, obs.shape[0]))
    if agent_types is not None:
        s["agent_types"] = np.asarray(agent_types, dtype=np.int32)
    return s


def advance_live_obstacles(obstacles, frequency=100.0):
    """dynamic_obstacle_node: cur_pos += cur_vel / frequency for the first M
    obstacles only (B/src/dynamic_obstacle_node.cpp:355-357); the trailing
    repulsive obstacle is not streamed."""
    out = obstacles.copy()
    out[:-1, 0:3] = out[:-1, 0:3] + out[:-1, 3:6] / frequency
    return out


# BASELINE.json configs as (N, H, M)
CONFIGS = {
    "C1": dict(n_agents=16, horizon=100, n_field=9),
    "C2": dict(n_agents=64, horizon=200, n_field=32),
    "C3": dict(n_agents=256, horizon=500, n_field=128),
    "C5": dict(n_agents=1024, horizon=200, n_field=32),
}


def config_scene(name, scene_id=0, dynamic=False):
    if name == "C1":
        return static1_scene(16, 100)
    c = CONFIGS[name]
    cid = {"C2": 2, "C3": 3, "C5": 5}[name]
    return synthetic_scene(c["n_agents"], c["horizon"], c["n_field"], cid, scene_id, dynamic)


def dual_arm_scenes(n_agents=256, horizon=200, n_field=32):
    """BASELINE config C4 (build-defined): one population per arm,
    2 x 256 agents, each arm's trailing repulsive obstacle = the other arm's end
    effector (shard.DualArmCoupling). The arms start 0.24 m apart and their
    goals are swapped in y, so the two self-collision spheres come into range
    mid-way. Returns [scene_arm0, scene_arm1]."""
    out = []
    for arm, (y0, y1) in enumerate(((-0.12, 0.10), (0.12, -0.10))):
        s = synthetic_scene(n_agents, horizon, n_field, 4, arm)
        s["start"] = np.array([-0.45, y0, 0.7])
        s["goal"] = np.array([0.45, y1, 0.7])
        s["name"] = "dual_arm_%d" % arm
        out.append(s)
    return out


def scene_from_record(rec, name="task", seed=0xC0FFEE00 + 9 * 256, horizon=None):
    """scene dict from a plain record (tests/golden/task_scenes.json: the
    reference's shipped task files reduced to planner inputs)"""
    obs = np.asarray(rec["obstacles"], dtype=np.float64)
    n = int(rec["n_agents"])
    s = dict(name=name, n_agents=n,
             max_prediction_steps=(horizon + 1) if horizon else int(rec["max_prediction_steps"]),
             dt=rec["dt"], velocity_max=rec["velocity_max"], approach_dist=rec["approach_dist"],
             detect_shell_rad=rec["detect_shell_rad"], agent_mass=1.0, radius=0.05,
             k_attr=rec["k_attr"], k_circ=rec["k_circ"], k_repel=rec["k_repel"], k_damp=rec["k_damp"],
             cost_gains=np.asarray(rec["cost_gains"], dtype=np.float64),
             ws_limits=np.asarray(rec["ws_limits"], dtype=np.float64),
             start=np.asarray(rec["start"], dtype=np.float64), goal=np.asarray(rec["goal"], dtype=np.float64),
             obstacles=obs)
    s["random_vecs"] = random_unit_vectors(SplitMix64(seed), n, obs.shape[0])
    return s
