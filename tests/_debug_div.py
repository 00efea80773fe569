import sys, numpy as np
sys.path.insert(0,'/root/repo')
import __graft_entry__ as g
pm=g.load_package()
from oracle.orc import OraclePlanner
np.set_printoptions(precision=17, linewidth=200)
sc=pm.scenes.static1_scene(10,1499)
hip=pm.PmafPlanner(sc,device=0,mgr_init_pos=sc["start"]); ora=OraclePlanner(sc,mgr_init_pos=sc["start"])
hip.set_initial_position(sc["start"]); ora.set_initial_position(sc["start"])
for t in range(6):
    bh=hip.tick(sc["obstacles"],sc["dt"],sc["cost_gains"],sc["ws_limits"]); bo=ora.tick(sc["obstacles"],sc["dt"],sc["cost_gains"],sc["ws_limits"])
    hip.stop()
    ph,nh=hip.paths(); po,no=ora.paths()
    d=np.abs(ph-po).max(axis=2)
    print("tick",t,"best",bh,bo,"n",nh,no)
    for a in range(10):
        bad=np.nonzero(d[a]>0)[0]
        if len(bad): print("  agent",a,"first diff step",bad[0],"diff there",d[a][bad[0]],"max",d[a].max(), "minobs",hip.min_obs_dist()[a],ora.min_obs_dist()[a])
    print("  minobs diff", np.abs(hip.min_obs_dist()-ora.min_obs_dist()).max(), "pl diff", np.abs(hip.path_lengths()-ora.path_lengths()).max())
