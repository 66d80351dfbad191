#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by scripts/profile_r6.sh) into the committed summaries under profiles/:
  profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats of the default bench run (search steps only)
  profiles/<tag>_bench_line.json      the bench line printed under the tracer (its HIP-event averages must agree)
  profiles/<tag>_pmc_<ctr>.csv        per-dispatch counters of the engine's kernels (FETCH_SIZE, WRITE_SIZE, LDS, TCC)
  profiles/<tag>_rocminfo.txt         the device the numbers were taken on
  profiles/traffic_r6.json            per kernel: HBM bytes per launch (+ LDS / L2 counters) keyed by the benched configuration —
                                      bench.py fills `roofline.traffic` from it only when the configuration matches
round 3 adds the SQ-wait / TA / TCP / TD counter groups (g1..g9) of the traversal kernel and their derived busy / stall fractions.
usage: scripts/summarize_profile_r5.py <tag>"""
import csv, json, os, shutil, statistics, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
# round 5: the headline traversal is the register-table bound form (graph_search_ubr_kernel) + its table kernel; a C5 run adds the robust prune
KEYS = {"gsearch_ubr": "graph_search_ubr_kernel", "ubr_table": "ubr_table_kernel", "gsearch": "graph_search_kernel", "exact_gather": "exact_gather_tr_kernel", "exact_gather_trq": "exact_gather_trq_kernel",
        "adc_mq": "adc_mq_kernel", "retain_diverse": "retain_diverse_kernel", "gsearch_pairc": "graph_search_pairc_kernel"}

for f, t in (("bench_kernel_stats.csv", f"{tag}_kernel_stats.csv"), ("rocminfo.txt", f"{tag}_rocminfo.txt")):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, t))
line = [l for l in open(os.path.join(src, "stats.log")) if l.startswith("{")] if os.path.exists(os.path.join(src, "stats.log")) else []
bench = json.loads(line[-1]) if line else {}
if line:
    open(os.path.join(dst, f"{tag}_bench_line.json"), "w").write(line[-1])


def rows_of(name):
    p = os.path.join(src, name)
    return list(csv.DictReader(open(p))) if os.path.exists(p) else []


def dur_ms(r):
    return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6


def steady(rows, kern):
    """dispatches of `kern` with the benched launch shape: the largest grid, and within it the longest-duration cluster
    (>= 70 % of the longest launch: the timed 65536-query batches, not the shorter calibration / evaluation / retry launches)"""
    d = [r for r in rows if kern in r["Kernel_Name"]]
    if not d:
        return []
    g = max(int(r["Grid_Size"]) for r in d)
    d = [r for r in d if int(r["Grid_Size"]) == g]
    names = {r["Counter_Name"] for r in d}
    one = [r for r in d if r["Counter_Name"] == sorted(names)[0]]
    top = max(dur_ms(r) for r in one)
    keep = {r["Dispatch_Id"] for r in one if dur_ms(r) >= 0.7 * top}
    return [r for r in d if r["Dispatch_Id"] in keep]


def write_pmc(rows, out):
    # only the dispatches of the priced kernels are kept (the full per-dispatch tables are tens of MB)
    rows = [r for r in rows if any(k in r["Kernel_Name"] for k in list(KEYS.values()) + ["exact_dense_kernel", "graph_search_wgx_kernel", "pair_scores", "adc_bq", "adc_kernel"])]
    if not rows:
        return
    with open(os.path.join(dst, out), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value", "duration_ms"])
        for r in rows:
            w.writerow([r["Kernel_Name"][:90], r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"],
                        r["Counter_Name"], r["Counter_Value"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6])


pmc = {c: rows_of(f"{c}_jv.csv") for c in ("FETCH_SIZE", "WRITE_SIZE", "LDS", "TCC")}
pmc_extra = {f"g{i}": rows_of(f"g{i}_jv.csv") for i in range(1, 10)}
pmc_extra.update({f"c5_g{i}": rows_of(f"c5_g{i}_jv.csv") for i in range(1, 10)})   # the same groups under `bench.py --workload c5` (robust prune, builder search)
pmc_wgx = {f"w{i}": rows_of(f"w{i}_jv.csv") for i in range(1, 10)}   # round 4: the same groups with the workgroup form forced
for gname, rows in pmc_wgx.items():
    write_pmc(rows, f"{tag}_pmc_{gname}.csv")
for gname, rows in pmc_extra.items():
    write_pmc(rows, f"{tag}_pmc_{gname}.csv")
for c, rows in pmc.items():
    write_pmc(rows, f"{tag}_pmc_{c.lower()}.csv")

# per-kernel duration from the kernel trace (untouched by counter collection)
trace = rows_of("kernel_trace_jv.csv")
cfg = bench.get("config", {})
entries = []
for key, kern in KEYS.items():
    e = {"kernel_key": key, "kernel": kern, "tag": tag}
    kcfg = {k: cfg.get(k) for k in ("n_vectors", "dim", "pq_subspaces", "queries_per_step", "rerankK")}
    if key == "adc_mq" and "flat_mode" in bench:
        kcfg.update(queries_per_step=bench["flat_mode"]["queries_per_step"], rerankK=bench["flat_mode"]["rerankK"])
    e["config"] = kcfg
    tr = [r for r in trace if kern in r["Kernel_Name"]]
    if tr:
        gs = lambda r: int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])  # noqa: E731
        g = max(gs(r) for r in tr)
        dur = [dur_ms(r) for r in tr if gs(r) == g]
        dur = [x for x in dur if x >= 0.7 * max(dur)]
        e["rocprof_avg_ms"], e["rocprof_launches"] = statistics.mean(dur), len(dur)
    def ctr(table, name):
        v = [float(r["Counter_Value"]) for r in steady(pmc[table], kern) if r["Counter_Name"] == name]
        return statistics.mean(v) if v else None
    fetch, write = ctr("FETCH_SIZE", "FETCH_SIZE"), ctr("WRITE_SIZE", "WRITE_SIZE")
    if fetch is not None and write is not None:
        # MI355X_MICROARCH.md §HBM: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts a 128-byte
        # request as 64 B for wide coalesced reads -> doubled (calibrated in round 1 on a known byte count); WRITE_SIZE as is
        e["fetch_size_raw_bytes"] = fetch * 1024
        e["fetch_bytes_corrected_x2"] = fetch * 2048
        e["write_bytes"] = write * 1024
        e["hbm_bytes_per_launch"] = fetch * 2048 + write * 1024
    for name in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_BUSY_CYCLES"):
        v = ctr("LDS", name)
        if v is not None:
            e[name] = v
    if e.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_fraction"] = e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"]
    # round 3: SQ wait / TA / TCP / TD counters of the traversal kernel (groups collected in their own --pmc passes)
    if key in ("gsearch", "gsearch_ubr", "ubr_table", "retain_diverse", "gsearch_pairc"):
        extra = {}
        for gname, rows in pmc_extra.items():
            for r in steady(rows, kern):
                extra.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for name, vals in sorted(extra.items()):
            e[name] = statistics.mean(vals)
        cu = 256.0
        if e.get("GRBM_GUI_ACTIVE"):
            cyc = e["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
            e["derived"] = {"cycles_per_launch": cyc,
                            "ta_busy_fraction": (e.get("TA_TA_BUSY_sum", 0.0) / cu) / cyc if e.get("TA_TA_BUSY_sum") else None,
                            "ta_addr_stalled_by_tc_fraction": (e.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum", 0.0) / cu) / cyc if e.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum") else None,
                            "td_busy_fraction": (e.get("TD_TD_BUSY_sum", 0.0) / cu) / cyc if e.get("TD_TD_BUSY_sum") else None,
                            "td_tc_stall_fraction": (e.get("TD_TC_STALL_sum", 0.0) / cu) / cyc if e.get("TD_TC_STALL_sum") else None,
                            "tcp_pending_stall_fraction": (e.get("TCP_PENDING_STALL_CYCLES_sum", 0.0) / cu) / cyc if e.get("TCP_PENDING_STALL_CYCLES_sum") else None,
                            "sq_wait_any_over_wave_cycles": e.get("SQ_WAIT_ANY", 0.0) / e["SQ_WAVE_CYCLES"] if e.get("SQ_WAVE_CYCLES") else None,
                            "sq_active_inst_over_wave_cycles": e.get("SQ_ACTIVE_INST_ANY", 0.0) / e["SQ_WAVE_CYCLES"] if e.get("SQ_WAVE_CYCLES") else None}
    hit, miss = ctr("TCC", "TCC_HIT_sum"), ctr("TCC", "TCC_MISS_sum")
    if hit is not None and miss is not None and hit + miss > 0:
        e["l2_hit_rate"] = hit / (hit + miss)
        e["TCC_HIT_sum"], e["TCC_MISS_sum"] = hit, miss
    if key == "gsearch" and any(pmc_wgx.values()):   # the workgroup form on the same batch: counters per launch
        w = {}
        for gname, rows in pmc_wgx.items():
            for r in steady(rows, "graph_search_wgx_kernel"):
                w.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        e["workgroup_form"] = {name: statistics.mean(v) for name, v in sorted(w.items())}
        durs = [dur_ms(r) for rows in pmc_wgx.values() for r in steady(rows, "graph_search_wgx_kernel")]
        if durs:
            e["workgroup_form"]["avg_ms_under_counters"] = statistics.mean(durs)
    if key == "exact_gather" and bench.get("roofline", {}).get("fused_rerank_rows"):
        continue   # (the fused rerank leaves this kernel the small launches only: no entry at the benched shape)
    if len(e) > 4:
        entries.append(e)

# ---- round 6: the two-stage flat filter at its three benched shapes (profile_r6.sh section B) ----
FLAT = {"c2": {"n_vectors": 1_000_000, "dim": 128, "pq_subspaces": 16, "queries_per_step": 1024},
        "c4": {"n_vectors": 12_500_000, "dim": 768, "pq_subspaces": 96, "queries_per_step": 256},
        "fm": {"n_vectors": 10_000_000, "dim": 768, "pq_subspaces": 96, "queries_per_step": 256}}


def last_line(name):
    p = os.path.join(src, name)
    if not os.path.exists(p):
        return {}
    l = [x for x in open(p) if x.startswith("{")]
    try:
        return json.loads(l[-1]) if l else {}
    except Exception:
        return {}


for w, kcfg in FLAT.items():
    line = last_line(f"{w}_trace.log") or last_line(f"{w}_FETCH_SIZE.log")
    rk = ((line.get("flat_mode") or (line.get("workloads") or {}).get("flat_mode") or {}).get("rerankK")) if w == "fm" else (line.get("config") or {}).get("rerankK")
    e = {"kernel_key": "adc_bq", "kernel": "adc_bq_kernel (+ adc_bq_table_kernel, the survivors' exact gather, adc_bq_count_kernel)", "tag": tag, "shape": w,
         "config": {**kcfg, "rerankK": rk}}

    def per_kernel(table, counter):
        rows = rows_of(f"{w}_{table}_jv.csv")
        out = {}
        for kern in ("adc_bq_kernel", "adc_bq_table_kernel", "adc_bq_count_kernel", "adc_kernel", "adc_gather"):
            d = [r for r in rows if kern in r["Kernel_Name"] and r["Counter_Name"] == counter]
            if not d:
                continue
            if kern == "adc_bq_kernel":   # the timed launches: the largest grid, the long-duration cluster
                d = [r for r in steady(rows, kern) if r["Counter_Name"] == counter]
            else:
                g = max(int(r["Grid_Size"]) for r in d)
                d = [r for r in d if int(r["Grid_Size"]) == g]
            out[kern] = statistics.mean(float(r["Counter_Value"]) for r in d)
        return out
    fetch, write = per_kernel("FETCH_SIZE", "FETCH_SIZE"), per_kernel("WRITE_SIZE", "WRITE_SIZE")
    if "adc_bq_kernel" in fetch:
        # MI355X_MICROARCH.md "HBM": KiB units; FETCH_SIZE tallies a 128-byte request as 64 B on gfx950 -> doubled; WRITE_SIZE as is
        e["per_kernel_hbm_bytes"] = {k: fetch.get(k, 0.0) * 2048 + write.get(k, 0.0) * 1024 for k in set(fetch) | set(write)}
        e["fetch_bytes_corrected_x2"] = fetch["adc_bq_kernel"] * 2048
        e["write_bytes"] = write.get("adc_bq_kernel", 0.0) * 1024
        e["hbm_bytes_per_launch"] = e["fetch_bytes_corrected_x2"] + e["write_bytes"]     # the bound scan alone (what roofline.traffic prices)
        e["hbm_bytes_per_launch_all_stages"] = sum(e["per_kernel_hbm_bytes"].values())
        e["hbm_compulsory_bytes"] = kcfg["n_vectors"] * (kcfg["pq_subspaces"] + 4)
        e["hbm_traffic_over_compulsory"] = e["hbm_bytes_per_launch"] / e["hbm_compulsory_bytes"]
    for name in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
        v = per_kernel("SQ_LDS_BANK_CONFLICT", name).get("adc_bq_kernel")
        if v is not None:
            e[name] = v
    if e.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_fraction"] = e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"]
    tr = rows_of(f"{w}_kernel_trace_jv.csv")
    durs = {}
    for r in tr:
        for kern in ("adc_bq_kernel", "adc_bq_table_kernel", "adc_bq_count_kernel", "adc_kernel", "topk", "exact_gather"):
            if kern in r["Kernel_Name"]:
                durs.setdefault(kern, []).append(dur_ms(r))
                break
    e["rocprof_kernel_ms"] = {k: {"launches": len(v), "avg_ms": statistics.mean(v), "max_ms": max(v)} for k, v in durs.items()}
    if "hbm_bytes_per_launch" in e or "SQ_LDS_IDX_ACTIVE" in e:   # (a pass that skipped section B leaves the earlier entries in place)
        entries.append(e)
        for f2 in ("FETCH_SIZE", "WRITE_SIZE", "SQ_LDS_BANK_CONFLICT"):
            write_pmc([r for r in rows_of(f"{w}_{f2}_jv.csv") if "adc" in r["Kernel_Name"]], f"{tag}_pmc_{w}_{f2.lower()}.csv")

# entries of earlier passes that this pass did not collect again stay (e.g. the flat filter's shapes when SKIP_FLAT=1 profiled the headline
# alone); an entry collected again replaces its predecessor — same kernel_key and shape
tj = os.path.join(dst, "traffic_r6.json")
if os.path.exists(tj):
    fresh = {(e["kernel_key"], e.get("shape")) for e in entries}
    entries += [e for e in json.load(open(tj)).get("entries", []) if (e["kernel_key"], e.get("shape")) not in fresh]
if bench.get("roofline", {}).get("fused_rerank_rows"):   # the traversal kernel of this pass also reranked (gs_body.h gs_rr_round)
    for e in entries:
        if e["kernel_key"] == "gsearch_ubr" and e["tag"] == tag:
            e["fused_rerank_rows"] = bench["roofline"]["fused_rerank_rows"]
out = {"source": f"rocprofv3 on `python bench.py` (default 10M workload, index cached so that only search steps are traced): "
                 f"scripts/profile_r6.sh <tag>; separate --pmc passes; profiles/<tag>_pmc_*.csv, profiles/<tag>_kernel_stats.csv; every entry names its tag",
       "entries": entries}
json.dump(out, open(os.path.join(dst, "traffic_r6.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
ks = os.path.join(dst, f"{tag}_kernel_stats.csv")
if os.path.exists(ks):
    for r in list(csv.DictReader(open(ks)))[:10]:
        print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
if bench:
    print("bench line under the tracer: value", bench.get("value"), "kernel_ms_per_step", bench.get("kernel_ms_per_step"))
