"""Where is the chip under-filled?  python tools/fill_timeline.py <kernel_trace.csv> [n_preprocess_per_step]
Takes the last step of a rocprofv3 kernel trace (multi-stream schedule), cuts it at every kernel start / end, and for every
slice adds up the workgroups of the kernels running in it (a kernel of G workgroups counts min(G, 256 * wg_per_cu) / (256 * wg_per_cu)
with wg_per_cu guessed from its LDS + workgroup size). Prints the time spent at fill < 0.25 / 0.5 / 1.0 and the kernels that own
the under-filled slices (longest first)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
npre = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pre = [i for i, r in enumerate(rows) if "preprocess" in r["Kernel_Name"]]
step = rows[pre[-npre - 1]:pre[-1]] if len(pre) > npre else rows


def cap(r):
    wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    lds = int(r.get("LDS_Block_Size", 0) or 0)
    per_cu = min(2048 // max(wg, 64), (160 * 1024) // lds if lds > 0 else 32, 16)
    vg = int(r.get("VGPR_Count", 0) or 0) + int(r.get("Accum_VGPR_Count", 0) or 0)
    if vg > 0:
        waves_simd = max(1, min(8, 512 // ((vg + 7) // 8 * 8)))
        per_cu = min(per_cu, max(1, waves_simd * 4 * 64 // max(wg, 64)))
    return 256 * max(per_cu, 1)


ev = []
for i, r in enumerate(step):
    g = (int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)) // max(1, int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1))
    f = min(1.0, g / cap(r))
    ev.append((int(r["Start_Timestamp"]), 1, i, f))
    ev.append((int(r["End_Timestamp"]), 0, i, f))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
live = {}
own = defaultdict(float)
hist = defaultdict(float)
prev = ev[0][0]
for t, kind, i, f in ev:
    dt = t - prev
    if dt > 0:
        fill = min(1.0, sum(live.values()))
        b = "idle" if not live else ("<0.25" if fill < 0.25 else "<0.5" if fill < 0.5 else "<1.0" if fill < 1.0 else "full")
        hist[b] += dt
        if fill < 0.5:
            names = sorted(live, key=lambda j: -live[j])
            key = " + ".join(step[j]["Kernel_Name"].split("(")[0][:40] for j in names[:2]) if names else "(nothing running)"
            own[key] += dt
    prev = t
    if kind == 1:
        live[i] = f
    else:
        live.pop(i, None)
span = (t1 - t0) / 1e6
print("step span %.3f ms, %d kernels" % (span, len(step)))
for b in ("idle", "<0.25", "<0.5", "<1.0", "full"):
    print("  fill %-6s %.3f ms" % (b, hist[b] / 1e6))
print("under-filled (< 0.5) time by what was running:")
for k, v in sorted(own.items(), key=lambda kv: -kv[1])[:30]:
    print("  %8.1f us  %s" % (v / 1e3, k))
