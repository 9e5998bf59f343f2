"""Condense a tools/collect_profiles.sh output directory into small committed artefacts under profiles/:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim, our kernels + top others)
   profiles/<tag>_pmc.json           per-kernel averages of the PMC passes, HBM traffic corrected per MI355X_MICROARCH.md
   profiles/<tag>_summary.md         human-readable table
"""
import collections, csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
what = sys.argv[3] if len(sys.argv) > 3 else "bench.py, X = 262144 x 8192 fp32, k = 64"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


stats = glob.glob(src + "/stats/**/*_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(stats[0]))) if stats else []
with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows[:24]:
        w.writerow([short(r["Name"])] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])

pmc = collections.defaultdict(dict)
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    fs = glob.glob(src + "/" + sub + "/**/*_counter_collection.csv", recursive=True)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(fs[0])):
        n = short(r["Kernel_Name"])
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[n][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for n, c in agg.items():
        if "kernel" not in n or "at::" in n:
            continue
        for k, v in c.items():
            pmc[n][k] = sum(v) / len(v)
        pmc[n]["dur_us_" + sub] = sum(dur[n].values()) / len(dur[n])
        pmc[n]["dispatches_" + sub] = len(dur[n])
for n, c in pmc.items():
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0                       # summed over the 8 XCDs
        # the quotient reads high on short dispatches (MI355X_MICROARCH.md, DVFS give-back: counter window > kernel time):
        # only reported for kernels of at least 50 us
        if c["dur_us_pmc_sq"] >= 50.0:
            c["clock_ghz"] = cyc / c["dur_us_pmc_sq"] / 1e3
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            c["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)   # 256 CUs x 4 SIMDs
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        # MI355X_MICROARCH.md (HBM): FETCH_SIZE is in KB and reports exactly half of the bytes of wide coalesced
        # streaming reads on gfx950 -> doubled; WRITE_SIZE (KB) is exact for 16-B-per-lane streaming stores.
        c["hbm_read_bytes"] = 2.0 * c.get("FETCH_SIZE", 0.0) * 1024.0
        c["hbm_write_bytes"] = c.get("WRITE_SIZE", 0.0) * 1024.0
        c["hbm_bytes"] = c["hbm_read_bytes"] + c["hbm_write_bytes"]
    if "TCC_HIT_sum" in c:
        c["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
json.dump(pmc, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)

with open(os.path.join(dst, tag + "_summary.md"), "w") as f:
    f.write("# rocprofv3 summary `%s` (`python3 %s`, 1 MI355X)\n\n" % (tag, what))
    so = os.path.join(src, "stdout.json")
    if os.path.exists(so):
        lines = [l for l in open(so).read().splitlines() if l.startswith("{")]
        if lines:
            f.write("Program output under the profiler (timings include its overhead):\n\n```\n%s\n```\n\n" % "\n".join(l[:1500] for l in lines[-6:]))
    f.write("## kernel-trace --stats (top kernels)\n\n| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n")
    for r in rows[:18]:
        f.write("| `%s` | %s | %.1f | %.2f | %s |\n" % (short(r["Name"])[:70], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                     float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    if any("clock_probe_kernel" in r["Name"] for r in rows):
        f.write("\n(`clock_probe_kernel` = bench.py's one-wave clock probe: it SLEEPS beside the steps it watches; its duration is the "
                "length of the watched window, not GPU work, and the percentages of the other rows are understated by its share.)\n")
    f.write("\n## PMC passes (separate runs; per-dispatch averages)\n\n| kernel | us (pmc) | clock GHz | MFMA busy | HBM read GB | HBM write GB | L2 hit |\n|---|---|---|---|---|---|---|\n")
    for n, c in sorted(pmc.items()):
        f.write("| `%s` | %.1f | %s | %s | %s | %s | %s |\n" % (
            n[:60], c.get("dur_us_pmc_sq", float("nan")),
            "%.2f" % c["clock_ghz"] if "clock_ghz" in c else "-",
            "%.1f%%" % (100 * c["mfma_busy_frac"]) if "mfma_busy_frac" in c else "-",
            "%.3f" % (c["hbm_read_bytes"] / 1e9) if "hbm_read_bytes" in c else "-",
            "%.3f" % (c["hbm_write_bytes"] / 1e9) if "hbm_write_bytes" in c else "-",
            "%.2f" % c["l2_hit_rate"] if "l2_hit_rate" in c else "-"))
print("profiles written for", tag)
