#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by scripts/profile_bench.sh) into the committed summaries under profiles/:
  profiles/<tag>_kernel_stats.csv         rocprofv3 --kernel-trace --stats (all kernels)
  profiles/<tag>_pmc_{fetch,write}.csv    per-dispatch FETCH_SIZE / WRITE_SIZE of the engine's kernels
  profiles/adc_traffic_r1.json            HBM bytes per launch of the dominant kernel (read by bench.py)
usage: scripts/summarize_profile.py <tag> <dominant-kernel-substring> [traffic-json-name]"""
import csv, json, os, shutil, sys
tag, dom = sys.argv[1], sys.argv[2]
traffic_name = sys.argv[3] if len(sys.argv) > 3 else "adc_traffic_r1.json"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "bench_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "stats.log")) if l.startswith("{")]
if line:
    open(os.path.join(dst, f"{tag}_bench_line.json"), "w").write(line[-1])
res = {}
for c, name in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    rows = list(csv.DictReader(open(os.path.join(src, f"{c}_jv.csv"))))
    with open(os.path.join(dst, f"{tag}_pmc_{name}.csv"), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value_KiB", "duration_ms"])
        for r in rows:
            w.writerow([r["Kernel_Name"][:90], r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"], r["Counter_Name"], r["Counter_Value"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6])
    d = [r for r in rows if dom in r["Kernel_Name"]]
    # the full-N launches are the ones with the largest grid
    g = max(int(r["Grid_Size"]) for r in d)
    vals = [float(r["Counter_Value"]) for r in d if int(r["Grid_Size"]) == g]
    vals = vals[len(vals) // 2:] if len(vals) > 6 else vals[-3:]   # steady-state launches
    res[name] = sum(vals) / len(vals) * 1024
    res[name + "_grid"] = g
fetch2 = res["fetch"] * 2  # MI355X_MICROARCH.md §HBM: gfx950 FETCH_SIZE counts 128-B requests as 64 B -> x2
json.dump({"kernel": dom, "tag": tag, "fetch_size_raw_bytes": res["fetch"], "fetch_bytes_corrected_x2": fetch2,
           "write_bytes": res["write"], "hbm_bytes_per_launch": fetch2 + res["write"], "grid_size": res["fetch_grid"],
           "correction": "FETCH_SIZE doubled per MI355X_MICROARCH.md (calibrated in r1 v0: a Q*N*4-byte coalesced read "
                         "reported exactly half); WRITE_SIZE used as is (KiB; calibrated: Q*N*4-byte store reported exactly)",
           "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; profiles/{tag}_pmc_*.csv"},
          open(os.path.join(dst, traffic_name), "w"), indent=1)
print(open(os.path.join(dst, traffic_name)).read())
for r in list(csv.DictReader(open(os.path.join(dst, f"{tag}_kernel_stats.csv"))))[:8]:
    print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
