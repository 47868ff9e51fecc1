"""Developer tool (GPU box): what the vendor's own fp64 GEMM (rocBLAS / hipBLASLt through torch.matmul) and its batched
Cholesky sustain on this chip with random operands -- an independent reading of the fp64 matrix-pipe ceiling quoted in
DESIGN.md section 3, and the vendor-library time for the per-ordering factorisation of a C3 step."""
import time, torch
dev = torch.device("cuda:0")
torch.manual_seed(0)

def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

for dt, name in ((torch.float64, "f64"), (torch.float32, "f32")):
    for n in (4096, 8192):
        a = torch.randn(n, n, dtype=dt, device=dev); b = torch.randn(n, n, dtype=dt, device=dev)
        for _ in range(3): a @ b
        s = timeit(lambda: a @ b, 10)
        print(f"gemm {name} {n}^3 random: {1e3*s:.2f} ms  {2*n**3/s/1e12:.1f} TFLOP/s", flush=True)
        z = torch.zeros_like(a)
        s = timeit(lambda: z @ z, 10)
        print(f"gemm {name} {n}^3 zeros : {1e3*s:.2f} ms  {2*n**3/s/1e12:.1f} TFLOP/s", flush=True)
    a = torch.randn(256, 1024, 1024, dtype=dt, device=dev); b = torch.randn(256, 1024, 1024, dtype=dt, device=dev)
    s = timeit(lambda: torch.bmm(a, b), 5)
    print(f"bmm {name} 256 x 1024^3 random: {1e3*s:.2f} ms  {256*2*1024**3/s/1e12:.1f} TFLOP/s", flush=True)
# vendor batched Cholesky of 512 SPD 1000 x 1000 matrices (= the two factorisations of a C3 step's 256 orderings)
x = torch.randn(512, 1000, 1200, dtype=torch.float64, device=dev)
spd = x @ x.transpose(1, 2) / 1200
del x
s = timeit(lambda: torch.linalg.cholesky(spd), 2)
print(f"torch.linalg.cholesky 512 x 1000^2 f64: {1e3*s:.1f} ms  {512*1000**3/3/s/1e12:.2f} TFLOP/s", flush=True)
L = torch.linalg.cholesky(spd[:256]); R = torch.linalg.cholesky(spd[256:])
s = timeit(lambda: torch.linalg.solve_triangular(L, R, upper=False), 2)
print(f"solve_triangular 256 x (1000^2 \\ 1000^2) f64: {1e3*s:.1f} ms  (dense rhs: {256*1000**3/s/1e12:.2f} TFLOP/s)", flush=True)
