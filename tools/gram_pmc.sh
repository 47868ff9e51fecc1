# FETCH_SIZE / WRITE_SIZE of the Gram kernels at the C3 shape, XCD-contiguous unit map (flags 0) against the natural one
# (flags 65536); tools/gram_time.py alternates them.  Separate passes per counter (MI355X_MICROARCH.md).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/gram_pmc; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 tools/gram_time.py ${GRAM_SHAPE:-c3} > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 tools/gram_time.py ${GRAM_SHAPE:-c3} > $O/w.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag, name in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    path = glob.glob(f"gpurun_out/gram_pmc/{tag}/**/*counter_collection.csv", recursive=True)[0]
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if "gram_kernel" not in r["Kernel_Name"]: continue
        d = rows.setdefault(int(r["Dispatch_Id"]), [0.0, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3])
        d[0] += float(r["Counter_Value"])
    vals = list(rows.values())
    print(name, "per gram_kernel dispatch (KB-units as reported, us):")
    for i, (v, us) in enumerate(vals): print(f"  #{i:2d} {v:14.1f}  {us:8.1f} us")
PY
