#!/usr/bin/env python3
"""Developer helper: per-kernel durations of the LAST occurrences in a rocprofv3 kernel trace (csv)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size','?'))}x{r.get('Grid_Size_Y','')} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size','?'))}  {r['Kernel_Name'][:90]}")
