import os, sys, importlib, time
os.environ["APDGICP_STATS"]="1"
sys.path.insert(0,'.')
import numpy as np, torch
reg = importlib.import_module("riv-slam_amd.registration"); scene = importlib.import_module("riv-slam_amd.scene")
import bench
P=8
b = reg.BatchAPDGICP(bench.bench_params(reg))
g=[]
for p in range(P):
    s,t,_,gs = scene.make_pair(8192,8192,scene.pair_seed(2,p),"odometry")
    b.add_cloud(s); b.add_cloud(t); g.append(gs)
r = b.align([(2*i,2*i+1) for i in range(P)], g)
st = b.debug_stats()
print("NN: groups/wave %.1f chunks tested/wave %.1f scanned/wave %.1f waves %d (per launch: 20 launches)" % (st[0]/st[3], st[1]/st[3], st[2]/st[3], st[3]))
print("KNN: groups loaded/wave %.1f  (query,group) pairs/wave %.1f  compactions/wave %.2f  waves %d" % (st[4]/st[7], st[9]/st[7], st[8]/st[7], st[7]))
