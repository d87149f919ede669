"""HBM-side traffic per launch of the conv kernels from two rocprofv3 --pmc passes over bench.py (FETCH_SIZE, WRITE_SIZE
in separate passes, MI355X_MICROARCH.md 'HBM'):  python3 tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
gfx950 corrections of that guide: FETCH_SIZE (KB) reports half of the bytes of wide (16 B/lane) streaming reads -> x2;
WRITE_SIZE (KB) is exact for 16 B/lane streaming stores. Infinity-Cache hits are counted (traffic past the L2)."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unit_amd.build import source_hash          # content hash of the kernel sources the profiled library was built from

by_grid = collections.defaultdict(lambda: collections.defaultdict(list))     # the grouped weight-gradient grids one by one (a Res5 head's is the large one)

def collect(d, counter):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                name = r["Kernel_Name"].split("(")[0].split("<")[0].strip()
                acc[name].append(float(r["Counter_Value"]))
                if "wgrad256_group" in name:
                    by_grid[int(r["Grid_Size"]) // int(r["Workgroup_Size"])][counter].append(float(r["Counter_Value"]))
    return acc

fetch = collect(sys.argv[1], "FETCH_SIZE")
write = collect(sys.argv[2], "WRITE_SIZE")
out = {"_note": "bytes past the L2 (HBM + Infinity Cache) per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over "
                "`bench.py --steps 2 --warmup 1 --no-overlap --no-roofline --no-cpu-baseline`; FETCH_SIZE doubled (gfx950 wide-read "
                "correction, MI355X_MICROARCH.md); KB = 1024 B",
       "_build_hash": source_hash()[:16]}          # bench.py prints `roofline.traffic` only while this equals the loaded library's stamp
groups = collections.defaultdict(lambda: [[], []])
for k in sorted(set(fetch) | set(write)):
    if not any(t in k for t in ("conv_igemm", "conv_wgrad", "multi_", "roi_align", "sgd")):
        continue
    # all launch variants of one kernel family (template instances, the halo7 variant) are one roofline line
    if "igemm256" in k:
        name = "conv_igemm256_kernel"
    elif "igemm_dma" in k and "ksplit" not in k:
        name = "conv_igemm_dma_kernel"
    elif "wgrad256" in k:
        name = "conv_wgrad256_kernel"
    else:
        name = k.split("_Z")[-1]
    groups[name][0] += fetch.get(k, [])
    groups[name][1] += write.get(k, [])
for name, (f, w) in groups.items():
    fb = 2.0 * 1024 * sum(f) / max(1, len(f)); wb = 1024.0 * sum(w) / max(1, len(w))
    out[name] = {"launches": max(len(f), len(w)), "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                 "hbm_bytes_per_launch": round(fb + wb)}
out["_conv_wgrad256_group_by_grid"] = {
    f"{wgs} workgroups": {"launches": len(c["FETCH_SIZE"]), "fetch_bytes_per_launch": round(2.0 * 1024 * sum(c["FETCH_SIZE"]) / max(1, len(c["FETCH_SIZE"]))),
                          "write_bytes_per_launch": round(1024.0 * sum(c["WRITE_SIZE"]) / max(1, len(c["WRITE_SIZE"])))}
    for wgs, c in sorted(by_grid.items())}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "igemm256" in k or "wgrad" in k}, indent=1))
