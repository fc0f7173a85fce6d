"""Condense rocprofv3 output of tools/profile_round.sh into small, commit-able summaries."""
import collections
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:70]


res = {"tag": tag}
def newest(pattern):
    """gpurun merges every call's output into the same directory: take the most recent file."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


stats = newest(f"{out}/trace/*/*kernel_stats.csv")
rows = []
if stats:
    for r in csv.DictReader(open(stats[0])):
        rows.append({"kernel": short(r["Name"]), "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
                     "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])})
res["kernel_stats"] = rows[:12]
for key, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = newest(f"{out}/pmc_{key}/*/*counter_collection.csv")
    agg = collections.defaultdict(lambda: [0.0, 0])
    if f:
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == cname:
                a = agg[short(r["Kernel_Name"])]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    res[key] = {k: {"sum_kb": v[0], "launches": v[1], "kb_per_launch": v[0] / max(v[1], 1)} for k, v in agg.items()
                if k.startswith("pio::")}
# HBM traffic per launch, gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE reports half of the
# bytes of wide (16 B/lane) coalesced reads -> x2; WRITE_SIZE is exact; both counters are in KiB.
traffic = {}
for k in res.get("fetch", {}):
    fk = res["fetch"][k]["kb_per_launch"]
    wk = res.get("write", {}).get(k, {}).get("kb_per_launch", 0.0)
    traffic[k] = {"bytes_per_launch": (2.0 * fk + wk) * 1024.0, "fetch_kb_raw": fk, "write_kb": wk}
# per kernel FAMILY (template instantiations merged, weighted by launches): what bench.py looks up by kernel name
fam = collections.defaultdict(lambda: [0.0, 0])
for k, t in traffic.items():
    n = res["fetch"][k]["launches"]
    f = fam[k.split("<")[0]]
    f[0] += t["bytes_per_launch"] * n
    f[1] += n
for k, (tot, n) in fam.items():
    if k not in traffic:
        traffic[k] = {"bytes_per_launch": tot / max(n, 1), "launches": n, "family_of": sorted(
            x for x in traffic if x.startswith(k + "<"))}
res["traffic"] = traffic
# Matrix-pipe occupancy per kernel from the counters: SQ_VALU_MFMA_BUSY_CYCLES (summed over every SIMD of the chip:
# 16 cycles per v_mfma_f32_16x16x32, 32 per 32x32x16) / (GRBM_GUI_ACTIVE / 8 XCDs = active cycles of the launch at the
# clock the chip held) / (256 CUs x 4 SIMDs).  This is occupancy AT THE HELD CLOCK; the fraction of the 2.5 PFLOP/s
# datasheet peak is lower by held clock / 2.4 GHz (MI355X_MICROARCH.md, DVFS give-back).
f = newest(f"{out}/pmc_mfma/*/*counter_collection.csv")
mf = collections.defaultdict(lambda: [0.0, 0.0, 0])
if f:
    for r in csv.DictReader(open(f[0])):
        a = mf[short(r["Kernel_Name"])]
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            a[0] += float(r["Counter_Value"])
            a[2] += 1
        elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            a[1] += float(r["Counter_Value"])
res["mfma_busy"] = {k: {"launches": v[2], "mfma_busy_cycles_per_launch": v[0] / max(v[2], 1),
                        "gui_active_cycles_per_launch_per_xcd": v[1] / 8.0 / max(v[2], 1),
                        "mfma_pipe_occupancy": v[0] / max(v[1] / 8.0 * 1024.0, 1.0)}
                    for k, v in mf.items() if k.startswith("pio::") and v[2]}
try:
    res["bench"] = json.loads(open(f"{out}/bench.json").read().strip().splitlines()[-1])
except Exception as e:  # noqa: BLE001
    res["bench"] = {"error": str(e)}
json.dump(res, open(f"{out}/summary_{tag}.json", "w"), indent=1)
print(json.dumps({"kernel_stats": res["kernel_stats"][:6], "traffic": traffic, "mfma_busy": res["mfma_busy"]},
                 indent=1)[:5000])
