"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected separately, as
MI355X_MICROARCH.md section HBM prescribes) into per-kernel HBM-side traffic per launch.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write <launches of last batch json>

Units and corrections: both counters are in KiB; on gfx950 FETCH_SIZE reads exactly 1/2 of the bytes
of a wide (16 B/lane) coalesced stream, so it is doubled; WRITE_SIZE is exact for 16-B/lane stores.
Only the LAST full batch of the run is used (the launches before it are warm-up / single orderings).
"""
import csv, glob, json, sys, collections

def load(d):
    f = sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True))[-1]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "lsspa" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("lsspa::", "")].append(float(r["Counter_Value"]))
    return agg

fetch, write = load(sys.argv[1]), load(sys.argv[2])
per_batch = {"gather_kernel": 1, "chol_diag2_kernel": 1, "chol_panel2_kernel": 7, "strip2_kernel": 1,
             "lift_partial_kernel": 1, "lift_finish_paired_kernel": 1, "stats_batch_kernel": 1}
rows, traffic = [], {}
for k, n in per_batch.items():
    f = sum(fetch[k][-n:]) * 1024.0
    w = sum(write[k][-n:]) * 1024.0
    hbm = 2.0 * f + w
    rows.append((k, n, f, 2.0 * f, w, hbm, hbm / n))
    traffic[k] = hbm / n
with open("profiles/r01_pmc_summary.csv", "w") as fh:
    fh.write("kernel,launches_per_batch,FETCH_SIZE_bytes_raw,fetch_bytes_corrected_x2,WRITE_SIZE_bytes,hbm_bytes_per_batch,hbm_bytes_per_launch\n")
    for r in rows:
        fh.write(",".join(str(x) for x in r) + "\n")
alias = {"gather": "gather_kernel", "chol_diag": "chol_diag2_kernel", "chol_panel": "chol_panel2_kernel",
         "strip": "strip2_kernel", "lift": "lift_partial_kernel"}
json.dump({"p": 1000, "batch_size": 128, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), "
           "tools/pmc_summary.py; FETCH_SIZE doubled (gfx950 16-B/lane correction)",
           "hbm_bytes_per_launch": {a: traffic[k] for a, k in alias.items()}}, open("profiles/pmc_traffic.json", "w"), indent=1)
for r in rows:
    print(f"{r[0]:22s} launches {r[1]:3d}  read {r[3]/1e9:7.2f} GB  write {r[4]/1e9:6.2f} GB  per launch {r[6]/1e9:6.2f} GB")
