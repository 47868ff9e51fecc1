"""Summarise tools/pmc_diag.sh: per kernel launch of the last profiled C3 step, counters side by side."""
import csv, collections, re, sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "pmc_diag")
def short(n): return re.sub(r"[<(].*", "", n).replace("void ", "").replace("lsspa::", "")
tables = {}
for tag in "abcde":
    path = os.path.join(ROOT, tag, f"{tag}_counter_collection.csv")
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if "lsspa" not in r["Kernel_Name"]: continue
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": short(r["Kernel_Name"]), "grid": int(r["Grid_Size"]),
                                                     "dur_us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    tables[tag] = list(disp.values())
# take the dispatches of the last full step: find the last gather_kernel
def last_step(rows):
    idx = max(i for i, r in enumerate(rows) if r["name"] == "gather_kernel")
    out = []
    for r in rows[idx:]:
        out.append(r)
        if r["name"].startswith("stats_merge"): break
    return out
steps = {t: last_step(rows) for t, rows in tables.items()}
n = min(len(s) for s in steps.values())
print(f"{'kernel':22s} {'grid':>9s} {'us':>8s} | clk GHz  MFMAbusy  CUbusy | waitLDS waitANY actVALU actLDS actVMEM lvlVMEM actANY bankconf | L2hit% DRAM/RD | VALU MFMA LDS SALU (M insts)")
for i in range(n):
    a, b, c, d, e = (steps[t][i] for t in "abcde")
    dur = a["dur_us"]
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0            # per XCD
    ghz = cyc / (dur * 1e3)
    wave_cyc = a["SQ_WAVE_CYCLES"]
    busycu = a["SQ_BUSY_CU_CYCLES"]
    mf = a["SQ_VALU_MFMA_BUSY_CYCLES"]
    # normalise the SQ cycle counters by total CU-cycles: 256 CUs x cycles (the counters sum over SEs/CUs; MFMA busy per SIMD x4)
    cu_cycles = 256.0 * cyc
    wb = b["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in b else None
    hit = c["TCC_HIT"] / max(1.0, c["TCC_HIT"] + c["TCC_MISS"])
    dram = c["TCC_EA0_RDREQ_DRAM"] / max(1.0, c["TCC_EA0_RDREQ"])
    print(f"{a['name']:22s} {a['grid']:9d} {dur:8.1f} | {ghz:5.2f}  {mf / (4 * cu_cycles):8.3f}  {busycu / cu_cycles:6.3f} | "
          f"{b['SQ_WAIT_INST_LDS'] / wave_cyc:7.3f} {b['SQ_WAIT_ANY'] / wave_cyc:7.3f} {b['SQ_ACTIVE_INST_VALU'] / (4*cu_cycles):7.3f} {b['SQ_ACTIVE_INST_LDS'] / (4*cu_cycles):6.3f} "
          f"{e['SQ_ACTIVE_INST_VMEM'] / (4*cu_cycles):7.3f} {e['SQ_INST_LEVEL_VMEM'] / (4*cu_cycles):7.2f} {e['SQ_ACTIVE_INST_ANY'] / (4*cu_cycles):6.3f} {e['SQ_LDS_BANK_CONFLICT'] / cu_cycles:8.3f} | "
          f"{100 * hit:6.1f} {dram:7.3f} | {d['SQ_INSTS_VALU'] / 1e6:6.1f} {d['SQ_INSTS_MFMA'] / 1e6:6.1f} {d['SQ_INSTS_LDS'] / 1e6:6.1f} {d['SQ_INSTS_SALU'] / 1e6:6.1f}")
