"""Print per-launch durations (us) of the Cholesky step kernels from a rocprofv3 kernel trace."""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
for key, n in (("chol_panel2", 7), ("chol_diag2", 1), ("chol_panel_", 15), ("strip", 1), ("gather", 1), ("lift_partial", 1)):
    lst = [r for r in rows if key in r["Kernel_Name"]][-n:]
    print(key, [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in lst])
