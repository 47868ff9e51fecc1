"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected separately, as
MI355X_MICROARCH.md section HBM prescribes) into per-kernel memory-side traffic per launch.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> \
        --label c3 --p 1000 --batch-size 128 --dtype f64

The profiled command must be `bench.py ... --no-probe --no-ttt --no-cpu-baseline` (every launch of a kernel then
belongs to a full-size step).  Units and corrections: both counters are in KiB; on gfx950 FETCH_SIZE reads exactly 1/2
of the bytes of a wide (16 B/lane) coalesced stream, so it is doubled; WRITE_SIZE is exact for 16-B/lane stores.
Infinity-Cache hits are counted (the counters sit on the L2's memory side), so this is fabric traffic, an upper bound
of the DRAM traffic.  Writes profiles/<tag>_pmc_summary_<label>.csv and merges the per-class figures into
profiles/pmc_traffic.json (read by bench.py for the `traffic` field of its roofline object).
"""
import argparse
import collections
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASS_OF = {"gather_kernel": "gather", "chol_diag2_kernel": "chol_diag", "chol_panel2_kernel": "chol_panel",
            "strip2_kernel": "strip", "lift_partial_kernel": "lift", "small_p_kernel": "small_p", "small_reg_kernel": "small_p",
            "gram_kernel": "gram"}


ONE_TIME = ("gram_kernel", "gram_reduce_kernel", "gram_finalize_kernel", "to_f32_kernel")   # per problem, not per step


def short(name):
    return re.sub(r"[<(].*", "", name).replace("void ", "").replace("lsspa::", "")


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "lsspa" in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch")
    ap.add_argument("write")
    ap.add_argument("--label", required=True)
    ap.add_argument("--p", type=int, required=True)
    ap.add_argument("--batch-size", type=int, required=True)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--tag", default="r03", help="round tag of the output file name")
    ap.add_argument("--steps", type=int, required=True, help="steps the profiled run executed (warm-up + timed + event pass)")
    args = ap.parse_args()
    fetch, write = load(args.fetch), load(args.write)
    rows, per_class, per_class_lps = [], {}, {}
    for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
        n = len(fetch[k])
        f_raw, w = sum(fetch[k]), sum(write.get(k, [0.0]))
        total = 2.0 * f_raw + w
        rows.append((k, n, f_raw, 2.0 * f_raw, w, total / n, 0.0 if k in ONE_TIME else total / args.steps))
        if k in CLASS_OF:
            per_class[CLASS_OF[k]] = total / n
            per_class_lps[CLASS_OF[k]] = n / args.steps
    out_csv = os.path.join(ROOT, "profiles", f"{args.tag}_pmc_summary_{args.label}.csv")
    with open(out_csv, "w") as fh:
        fh.write("kernel,launches,FETCH_SIZE_bytes_raw,fetch_bytes_corrected_x2,WRITE_SIZE_bytes,"
                 "bytes_per_launch,bytes_per_step (0 for the once-per-problem reduction kernels)\n")
        for r in rows:
            fh.write(",".join(str(x) for x in r) + "\n")
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    doc = {"runs": []}
    if os.path.exists(path):
        try:
            old = json.load(open(path))
            doc["runs"] = [r for r in old.get("runs", []) if r.get("label") != args.label]
        except Exception:
            pass
    doc["runs"].append({"label": args.label, "p": args.p, "batch_size": args.batch_size, "dtype": args.dtype,
                        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py "
                                  "--no-probe --no-ttt --no-cpu-baseline; tools/pmc_summary.py; FETCH_SIZE doubled "
                                  "(gfx950 16-B/lane correction); fabric-side bytes, Infinity-Cache hits included",
                        "hbm_bytes_per_launch": per_class,
                        "launches_per_step": per_class_lps,
                        "bytes_per_step": sum(r[6] for r in rows)})
    json.dump(doc, open(path, "w"), indent=1)
    for r in rows:
        print(f"{r[0]:28s} launches {r[1]:4d}  read {r[3] / r[1] / 1e9:8.3f} GB  write {r[4] / r[1] / 1e9:7.3f} GB per launch"
              f"   {r[6] / 1e9:8.3f} GB per step")
    print(f"total per step: {sum(r[6] for r in rows) / 1e9:.2f} GB")


if __name__ == "__main__":
    main()
