"""Summarise a rocprofv3 --pmc run directory: per kernel, average counter values and derived MFMA utilisation / clock."""
import collections, csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
seen = set()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    key = (r["Dispatch_Id"])
    if key not in seen:
        seen.add(key)
        dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, c in agg.items():
    if not any(x in n for x in (sys.argv[2:] or ["nt_kernel", "tn_kernel", "sqnorm"])):
        continue
    avg = {k: sum(v) / len(v) for k, v in c.items()}
    us = sum(dur[n][1:] or dur[n]) / max(1, len(dur[n][1:] or dur[n]))
    line = "%-50s n=%3d dur=%9.1fus" % (n, len(dur[n]), us)
    if "GRBM_GUI_ACTIVE" in avg:
        cyc = avg["GRBM_GUI_ACTIVE"] / 8
        line += " clk=%.2fGHz" % (cyc / us / 1e3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
            line += " mfma_busy=%.1f%%" % (100 * avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc))
    for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_INST_CYCLES_VMEM_RD", "TCP_PENDING_STALL_CYCLES_sum", "TA_BUSY_avr", "TCP_TCC_READ_REQ_sum"):
        if k in avg:
            line += " %s=%.4g" % (k, avg[k])
    print(line)
