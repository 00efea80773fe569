"""MI355X-native predictive multi-agent circular-field planner tick.

The directory name carries hyphens (it is named after the reference
repository), so import it through `__graft_entry__.load_package()`, which
registers it as the module `pmaf_amd`:

    import __graft_entry__ as g
    pmaf = g.load_package()
    planner = pmaf.PmafPlanner(pmaf.scenes.config_scene("C2"))

Contents: csrc/ (HIP kernels + C-ABI, built into lib/libpmaf_hip.so),
planner.py (ctypes binding of include/pmaf.h), scenes.py (task scenes and the
seeded synthetic scenes), shard.py (population sharding across ranks: the
orchestration over the C-ABI's communicator / winner-exchange entry points).
"""
from . import scenes, shard  # noqa: F401
from .planner import (LIB_PATH, SYMBOLS, PmafComm, PmafError, PmafPlanner, debug_math, device_count,  # noqa: F401
                      load_library, select_best)  # noqa: F401
