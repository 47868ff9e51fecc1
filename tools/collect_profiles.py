"""Copy the artefacts of tools/r02_final.sh from gpurun_out/ (scratch) into profiles/ (tracked) and derive the
PMC traffic summaries.  Run in the build container after the gpurun call has merged its output."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r02_final")
DST = os.path.join(ROOT, "profiles")
os.makedirs(DST, exist_ok=True)


def cp(src, dst):
    shutil.copyfile(os.path.join(SRC, src), os.path.join(DST, dst))
    print("profiles/" + dst)


for name in ("bench_c3", "bench_c2", "bench_c5", "bench_c3_f32", "bench_c3_rehearse", "bench_c3_strong16_rehearse"):
    cp(name + ".json", "r02_" + name + ".json")
with open(os.path.join(DST, "r02_mfma_bench5.log"), "w") as out:
    out.write("# tools/mfma_bench5.hip on the MI355X of this run: the k-loop of the two-level kernels in isolation\n"
              "# (global prefetch -> LDS -> barrier -> fragments -> MFMA, 128 x 128 tile per workgroup, 16 flop per operand byte)\n\n"
              "## all-zero operands\n" + open(os.path.join(SRC, "mfma_bench5_zero.log")).read() +
              "\n## random operands (./mfma_bench5 random)\n" + open(os.path.join(SRC, "mfma_bench5_random.log")).read())
print("profiles/r02_mfma_bench5.log")
with open(os.path.join(DST, "r02_kloop_ceiling.log"), "w") as out:
    out.write("# What the fp64 / fp32 matrix pipe delivers on the MI355X of this run (tools/r02_final.sh)\n\n"
              "## tools/mfma_bench6.hip: the fp64 k-loop of the two-level kernels in isolation, steady-state clock\n" +
              open(os.path.join(SRC, "mfma_bench6.log")).read() +
              "\n## tools/mfma_bench7.hip: the fp32 k-loop\n" + open(os.path.join(SRC, "mfma_bench7.log")).read() +
              "\n## tools/dgemm_probe.py: the vendor library's GEMM / batched GEMM / batched Cholesky / triangular solve (torch -> rocBLAS / hipBLASLt / rocSOLVER)\n" +
              "".join(l for l in open(os.path.join(SRC, "dgemm_probe.log")) if "amdgpu.ids" not in l) +
              "\n## tools/dgemm_probe2.py: shapes like ours (X^T X at C3, batched 1024^3 NN / NT)\n" +
              "".join(l for l in open(os.path.join(SRC, "dgemm_probe2.log")) if "amdgpu.ids" not in l) +
              "\n## tools/gram_time.py: our Gram reduction, both sides per line (C3: 2 x 1.003e11 flop; C5 per side as printed)\n" +
              "".join(l for l in open(os.path.join(SRC, "gram_time.log")) if "amdgpu.ids" not in l))
print("profiles/r02_kloop_ceiling.log")
# (config, p, batch, dtype, steps executed by the PMC runs = warm-up + timed + event pass)
for cfg, p, b, dt, steps in (("c3", 1000, 128, "f64", 2 + 6 + 6), ("c5", 5000, 128, "f32", 1 + 3 + 3),
                             ("c2", 100, 128, "f64", 8 + 16 + 16)):
    cp(f"stats_{cfg}/{cfg}_kernel_stats.csv", f"r02_{cfg}_kernel_stats.csv")
    cp(f"stats_{cfg}.json", f"r02_{cfg}_bench_under_rocprof.json")
    cp(f"fetch_{cfg}/{cfg}_counter_collection.csv", f"r02_{cfg}_pmc_fetch_size.csv")
    cp(f"write_{cfg}/{cfg}_counter_collection.csv", f"r02_{cfg}_pmc_write_size.csv")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"),
                    os.path.join(DST, f"r02_{cfg}_pmc_fetch_size.csv"), os.path.join(DST, f"r02_{cfg}_pmc_write_size.csv"),
                    "--label", cfg, "--p", str(p), "--batch-size", str(b), "--dtype", dt, "--steps", str(steps)], check=True)
