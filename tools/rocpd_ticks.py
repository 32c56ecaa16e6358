#!/usr/bin/env python3
"""Durations of the 20 search and 20 linearize launches of the LAST align of a kernel trace (tools/one_batch.py alone on the GPU).
usage: rocpd_ticks.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
# last align: last 20 nn + 20 lin
nn = [(s, e) for n, s, e in rows if 'k_nn_' in n][-20:]
li = [(s, e) for n, s, e in rows if 'k_linearize' in n][-20:]
print("nn  us:", " ".join(f"{(e - s) / 1e3:.0f}" for s, e in nn))
print("lin us:", " ".join(f"{(e - s) / 1e3:.0f}" for s, e in li))
kn = [(n.split('(')[0][-22:], (e - s) / 1e3) for n, s, e in rows if 'knn' in n or 'sort' in n or 'merge' in n or 'boxes' in n or 'regular' in n or 'pack' in n][-6:]
print(kn)
