#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory (gpurun_out/prof_<tag>/) into small, committed
summaries under profiles/: <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, top kernels),
<tag>_pmc.json (per-kernel averages of every collected counter + derived HBM traffic per launch)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)


def short(name):
    name = name.replace("void ", "")
    return name if len(name) < 110 else name[:107] + "..."


def newest(pattern):
    """gpurun merges every call's files into the same local directory (one <pid>_ prefix per run): the newest only."""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:] if fs else []


stats = newest(f"{src}/trace/runc/*_kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:10]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"]])

pmc = collections.defaultdict(dict)
for f in [g for d in sorted(glob.glob(f"{src}/pmc_*")) for g in newest(f"{d}/runc/*_counter_collection.csv")]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    clk = collections.defaultdict(list)               # kernel -> [(GRBM_GUI_ACTIVE, dispatch duration in ns)]
    for r in csv.DictReader(open(f)):
        if "drrt::" not in r["Kernel_Name"]:
            continue
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[kn]["_VGPR"] = [float(r["VGPR_Count"])]
        agg[kn]["_LDS_bytes_per_block"] = [float(r["LDS_Block_Size"])]
        dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        agg[kn]["_dispatch_ns_under_pmc"].append(dur)
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            clk[kn].append((float(r["Counter_Value"]), dur))
    for kn, c in agg.items():
        for k, v in c.items():
            pmc[kn][k] = sum(v) / len(v)
    for kn, v in clk.items():
        # effective shader clock = GRBM_GUI_ACTIVE (sum over the 8 XCDs) / 8 / dispatch duration; dispatches that return at
        # once (the adjoint kernel the bundle classification did not pick) read high and are left out
        dmax = max(d for _, d in v)
        live = [(c_, d) for c_, d in v if d >= 0.5 * dmax and d > 0]
        if live:
            pmc[kn]["effective_clock_ghz"] = sum(c_ / 8.0 / d for c_, d in live) / len(live)
            pmc[kn]["effective_clock_dispatch_ns"] = sum(d for _, d in live) / len(live)
for kn, c in list(pmc.items()):
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE
        # counts 64 B per 128-B request for wide coalesced streams (x2 correction); scattered dword gathers are
        # uncalibrated, so both the raw and the x2-corrected read figure are kept.
        c["hbm_read_bytes_raw"] = c["FETCH_SIZE"] * 1024
        c["hbm_read_bytes_x2"] = c["FETCH_SIZE"] * 2048
        c["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
        c["hbm_traffic_bytes_per_launch"] = c["hbm_read_bytes_x2"] + c["hbm_write_bytes"]
    if "SQ_WAIT_INST_ANY" in c and c.get("SQ_WAVE_CYCLES"):
        c["wait_frac"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]        # share of wave-cycles spent waiting on anything
meta = {"tag": tag}
for b in glob.glob(f"{src}/bench_trace.json"):
    open(f"profiles/{tag}_bench_under_rocprof.json", "w").write(open(b).read())
    try:                              # which library produced these counters (drrt_version(): digest of its sources)
        meta["lib_version"] = json.loads(open(b).read().strip().splitlines()[-1]).get("lib_version")
    except Exception:
        pass
pmc["_meta"] = meta
json.dump(pmc, open(f"profiles/{tag}_pmc.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: {kk: round(vv, 1) for kk, vv in v.items() if kk.startswith("hbm") or kk in ("FETCH_SIZE", "WRITE_SIZE")}
                  for k, v in pmc.items() if k != "_meta"}, indent=1))
