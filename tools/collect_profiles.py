"""Copy the artefacts of tools/r0N_final.sh (python tools/collect_profiles.py r04) from gpurun_out/ (scratch) into profiles/ (tracked) and derive the
PMC traffic summaries.  Run in the build container after the gpurun call has merged its output."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"      # round tag: gpurun_out/<tag>_final -> profiles/<tag>_*
SRC = os.path.join(ROOT, "gpurun_out", f"{TAG}_final")
DST = os.path.join(ROOT, "profiles")
os.makedirs(DST, exist_ok=True)


def cp(src, dst):
    shutil.copyfile(os.path.join(SRC, src), os.path.join(DST, dst))
    print("profiles/" + dst)


for name in ("bench_c3", "bench_c3_correlated", "bench_c2", "bench_c5", "bench_c3_f32", "bench_c3_rehearse",
             "bench_c3_strong16_rehearse", "bench_c3_one_lane", "bench_c3_80steps", "bench_c3_two_ranks_one_gpu"):
    if os.path.exists(os.path.join(SRC, name + ".json")):
        cp(name + ".json", f"{TAG}_" + name + ".json")
with open(os.path.join(DST, f"{TAG}_probes.log"), "w") as out:
    for title, f in (("tools/gram_time.py: the Gram reduction, both sides per line (C3: 2 x 1.003e11 flop; C5 per side as printed); "
                      "flags 0 = XCD-contiguous unit map, 65536 = natural map", "gram_time.log"),
                     ("tools/gram_pmc.sh: FETCH_SIZE / WRITE_SIZE (KiB as reported; FETCH to be doubled) per gram_kernel dispatch, "
                      "the dispatches alternating between the two maps as in gram_time.py", "gram_pmc.log"),
                     ("tools/lat_probe.hip: dependent-chain latencies on the 16 x 16 elimination's critical path", "lat_probe.log"),
                     ("tools/small_probe.hip: phases of the fused small-p kernel (p = 100, 2048 orderings)", "small_probe.log"),
                     ("tools/factor_probe.hip: phases of the 64 x 64 / 128 x 128 diagonal factorisations", "factor_probe.log"),
                     ("tools/xtile_probe.hip: every 61st workgroup of the eight panel launches of a C3 step (L tiles, X tiles): "
                      "start time in its launch and phase durations, us", "xtile_probe.log"),
                     ("tools/host_reduce_probe.py: the streamed reduction from host arrays at the C3 shape", "host_reduce_probe.log")):
        if not os.path.exists(os.path.join(SRC, f)):
            continue
        out.write(f"## {title}\n" + "".join(l for l in open(os.path.join(SRC, f)) if "amdgpu.ids" not in l) + "\n")
print(f"profiles/{TAG}_probes.log")
if os.path.exists(os.path.join(SRC, "pmc_diag_summary.txt")):
    shutil.copyfile(os.path.join(SRC, "pmc_diag_summary.txt"), os.path.join(DST, f"{TAG}_c3_pmc_diag.txt"))
    print(f"profiles/{TAG}_c3_pmc_diag.txt")
# (config, p, batch, dtype, steps executed by the PMC runs = warm-up + timed + event pass)
for cfg, p, b, dt, steps in (("c3", 1000, 128, "f64", 2 + 6 + 6), ("c5", 5000, 128, "f32", 1 + 3 + 3),
                             ("c2", 100, 128, "f64", 64)):   # (warm-up + timed + event pass; no sustained region: $P;
                                                             #  c2: four launches of a 16-step group -- a priming launch
                                                             #  (bench.py: the group's workspace before anything is timed),
                                                             #  warm-up, timed region, event pass)
    if not os.path.exists(os.path.join(SRC, f"stats_{cfg}", f"{cfg}_kernel_stats.csv")):
        print(f"(no {cfg} artefacts in {SRC})")
        continue
    cp(f"stats_{cfg}/{cfg}_kernel_stats.csv", f"{TAG}_{cfg}_kernel_stats.csv")
    cp(f"stats_{cfg}.json", f"{TAG}_{cfg}_bench_under_rocprof.json")
    if os.path.exists(os.path.join(SRC, f"stats_{cfg}l1", f"{cfg}l1_kernel_stats.csv")):     # the same command with --lanes 1
        cp(f"stats_{cfg}l1/{cfg}l1_kernel_stats.csv", f"{TAG}_{cfg}_one_lane_kernel_stats.csv")
        cp(f"stats_{cfg}l1.json", f"{TAG}_{cfg}_one_lane_bench_under_rocprof.json")
    cp(f"fetch_{cfg}/{cfg}_counter_collection.csv", f"{TAG}_{cfg}_pmc_fetch_size.csv")
    cp(f"write_{cfg}/{cfg}_counter_collection.csv", f"{TAG}_{cfg}_pmc_write_size.csv")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"),
                    os.path.join(DST, f"{TAG}_{cfg}_pmc_fetch_size.csv"), os.path.join(DST, f"{TAG}_{cfg}_pmc_write_size.csv"),
                    "--label", cfg, "--p", str(p), "--batch-size", str(b), "--dtype", dt, "--steps", str(steps), "--tag", TAG],
                   check=True)

# the public call over a many-check run (tools/full_run_probe.py), alone and under rocprofv3 --kernel-trace --stats
with open(os.path.join(DST, f"{TAG}_full_run_probe.log"), "w") as out:
    for title, f in (("tools/full_run_probe.py 100 10000 64 (C2: the public call over 65 checks)", "full_run_c2.log"),
                     ("tools/full_run_probe.py 1000 100000 128 (C3: 129 checks)", "full_run_c3.log"),
                     (f"the same under rocprofv3 --kernel-trace --stats (profiles/{TAG}_full_run_c2_kernel_stats.csv)", "stats_fr2.log"),
                     (f"the same under rocprofv3 --kernel-trace --stats (profiles/{TAG}_full_run_c3_kernel_stats.csv)", "stats_fr3.log")):
        if os.path.exists(os.path.join(SRC, f)):
            out.write(f"## {title}\n" + "".join(l for l in open(os.path.join(SRC, f)) if l.startswith("{")) + "\n")
print(f"profiles/{TAG}_full_run_probe.log")
for cfg in ("fr2", "fr3"):
    f = os.path.join(SRC, f"stats_{cfg}", f"{cfg}_kernel_stats.csv")
    if os.path.exists(f):
        cp(f"stats_{cfg}/{cfg}_kernel_stats.csv", f"{TAG}_full_run_c{cfg[-1]}_kernel_stats.csv")
