"""time per step during which the kernels in flight have fewer than 256 workgroups between them (one per CU): tools/underfill_stats.py <kernel_trace.csv>
A coarse picture of where a step leaves the chip empty; segments listed for the last step."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['wg'] = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // (int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']))
rows.sort(key=lambda r: r['s'])
pre = [i for i, r in enumerate(rows) if "preprocess_kernel" in r["Kernel_Name"]]
per = 4
TH = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nsteps = 6
bounds = [pre[-per * k - 1] for k in range(nsteps, -1, -1)]
tot_under = tot_span = 0
for a, b in zip(bounds[:-1], bounds[1:]):
    step = rows[a:b]
    ev = []
    for r in step:
        ev.append((r['s'], r['wg'])); ev.append((r['e'], -r['wg']))
    ev.sort()
    live = 0; prev = ev[0][0]; under = 0; segs = []
    for t, d in ev:
        if live < TH and t > prev:
            under += t - prev
            if segs and segs[-1][1] == prev: segs[-1][1] = t
            else: segs.append([prev, t])
        prev = t; live += d
    span = ev[-1][0] - ev[0][0]
    tot_under += under; tot_span += span
print(f"per step: span {tot_span / nsteps / 1e6:.3f} ms, < {TH} workgroups in flight for {tot_under / nsteps / 1e6:.3f} ms")
t0 = ev[0][0]
print("last step, segments > 30 us (start, length in us):", [(round((a - t0) / 1e3), round((b - a) / 1e3)) for a, b in segs if b - a > 30000])
