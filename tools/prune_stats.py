"""Pruning statistics of the NN and k-NN kernels (APDGICP_STATS): python tools/prune_stats.py [max_iterations]
env P (pairs, default 8), NS / NT (points per source / target cloud, default 8192)"""
import os, sys, importlib, time
os.environ["APDGICP_STATS"]="1"
sys.path.insert(0,'.')
import numpy as np, torch
reg = importlib.import_module("riv-slam_amd.registration"); scene = importlib.import_module("riv-slam_amd.scene")
import bench
P=int(os.environ.get('P','8'))
NS=int(os.environ.get('NS','8192')); NT=int(os.environ.get('NT','8192'))
prm = bench.bench_params(reg)
if len(sys.argv) > 1: prm.max_iterations = int(sys.argv[1])
b = reg.BatchAPDGICP(prm)
g=[]
for p in range(P):
    s,t,_,gs = scene.make_pair(NS,NT,scene.pair_seed(2,p),"odometry")
    b.add_cloud(s); b.add_cloud(t); g.append(gs)
b.compute_covariances(); kst = b.debug_stats()
r = b.align([(2*i,2*i+1) for i in range(P)], g)
st = b.debug_stats()
print("NN ticks/wave (sampled): start+hint %.0f  masks %.0f  groups %.0f" % (st[10]/st[14], st[11]/st[14], st[12]/st[14]))
print("NN: box batches/wave %.1f groups/wave %.1f chunks tested/wave %.1f scanned/wave %.1f waves %d (%d launches)" % (st[5]/st[3], st[0]/st[3], st[1]/st[3], st[2]/st[3], st[3], prm.max_iterations))
st = kst
print("KNN: groups loaded/wave %.1f  (query,group) pairs/wave %.1f  compactions/wave %.2f  waves %d" % (st[4]/st[7], st[9]/st[7], st[8]/st[7], st[7]))
