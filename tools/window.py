"""print the kernel-trace timeline of the LAST step around an anchor kernel: tools/window.py <trace.csv> <anchor substring> <before us> <after us> [min_dur us]"""
import csv, sys
path, anchor, before, after = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4])
mind = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
which = int(sys.argv[6]) if len(sys.argv) > 6 else -1
rows = list(csv.DictReader(open(path)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
t0 = rows[idx[which]]['s']
for r in rows:
    if r['e'] > t0 - before * 1e3 and r['s'] < t0 + after * 1e3 and (r['e'] - r['s']) / 1e3 >= mind:
        print(f"{(r['s']-t0)/1e3:9.1f} {(r['e']-r['s'])/1e3:8.1f} q{r['Queue_Id']} s{r['Stream_Id']} g{int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d} {r['Kernel_Name'][:90]}")
