#!/usr/bin/env python3
"""Median duration per (kernel, grid) from a rocprofv3 rocpd database: kernel_medians.py <dir or .db> [name filter]"""
import sqlite3, glob, sys, os, collections
path = sys.argv[1]
dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for db in dbs:
    c = sqlite3.connect(db)
    d = collections.defaultdict(list)
    for name, st, en, gx, gy in c.execute("select name, start, end, grid_x, grid_y from kernels"):
        if flt in name: d[(name.replace("void asg::(anonymous namespace)::", "")[:60], gx, gy)].append((en - st) / 1000)
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:12]:
        v = sorted(v)
        print("%-62s grid %8d,%2d n=%3d median %8.1f us min %8.1f" % (k[0], k[1], k[2], len(v), v[len(v) // 2], v[0]))
