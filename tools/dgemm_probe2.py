"""Developer tool: fp64 GEMM shapes like ours through the vendor library, for a rocprofv3 kernel trace (kernel names
encode the macro tile / wave tile the library picked)."""
import time, torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
dt = torch.float64
for (m, n, k) in ((8192, 8192, 8192), (4096, 4096, 4096), (1024, 1024, 100000)):
    a = torch.randn(m, k, dtype=dt, device=dev); b = torch.randn(k, n, dtype=dt, device=dev)
    s = timeit(lambda: a @ b, 5)
    print(f"gemm f64 {m}x{n}x{k} NN random: {1e3*s:.2f} ms  {2*m*n*k/s/1e12:.1f} TFLOP/s", flush=True)
    if k == 100000:
        at = a.t().contiguous()      # [k][m]: the Gram shape, X^T X with X row-major
        s = timeit(lambda: at.t() @ b, 5)
        print(f"gemm f64 {m}x{n}x{k} TN (X^T X layout) random: {1e3*s:.2f} ms  {2*m*n*k/s/1e12:.1f} TFLOP/s", flush=True)
a = torch.randn(256, 1024, 1024, dtype=dt, device=dev); b = torch.randn(256, 1024, 1024, dtype=dt, device=dev)
s = timeit(lambda: torch.bmm(a, b), 5)
print(f"bmm f64 256 x 1024^3 NN random: {1e3*s:.2f} ms  {256*2*1024**3/s/1e12:.1f} TFLOP/s", flush=True)
s = timeit(lambda: torch.bmm(a, b.transpose(1, 2)), 5)
print(f"bmm f64 256 x 1024^3 NT random: {1e3*s:.2f} ms  {256*2*1024**3/s/1e12:.1f} TFLOP/s", flush=True)
