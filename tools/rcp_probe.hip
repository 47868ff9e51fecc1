// Accuracy of the hardware reciprocal / reciprocal square root and of one and two Newton steps on them (developer tool;
// tiles.h: fast_recip, fast_rsqrt).  Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/rcp_probe tools/rcp_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* d, double* r0, double* r1, double* r2, double* s0, double* s1, double* s2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  double x = d[i];
  double r = __builtin_amdgcn_rcp(x); r0[i] = r;
  r = fma(fma(-x, r, 1.0), r, r); r1[i] = r;
  r = fma(fma(-x, r, 1.0), r, r); r2[i] = r;
  double q = __builtin_amdgcn_rsq(x); s0[i] = q;
  q = fma(0.5 * q, fma(-x * q, q, 1.0), q); s1[i] = q;
  q = fma(0.5 * q, fma(-x * q, q, 1.0), q); s2[i] = q;
}
int main() {
  const int n = 1 << 20; double* h = new double[n];
  unsigned long long z = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; h[i] = ldexp(1.0 + (z >> 12) * (1.0 / 4503599627370496.0), (int)(z % 40) - 20); }
  double *d, *o[6]; hipMalloc(&d, n * 8); for (auto& p : o) hipMalloc(&p, n * 8);
  hipMemcpy(d, h, n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o[0], o[1], o[2], o[3], o[4], o[5], n);
  double* r = new double[n];
  const char* nm[6] = {"rcp", "rcp+1", "rcp+2", "rsq", "rsq+1", "rsq+2"};
  for (int j = 0; j < 6; ++j) {
    hipMemcpy(r, o[j], n * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < n; ++i) {
      long double ex = j < 3 ? 1.0L / h[i] : 1.0L / sqrtl((long double)h[i]);
      double e = fabs((double)((r[i] - ex) / ex)); if (e > worst) worst = e;
    }
    printf("%-6s max relative error %.3e (%.2f ulp of 2^-52)\n", nm[j], worst, worst / 2.220446049250313e-16);
  }
  return 0;
}
