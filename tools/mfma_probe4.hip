// Probe the lane maps of v_mfma_f64_4x4x4_4b_f64 (with and without A-block broadcast):
// for every pair (la, lb) run the instruction with one-hot A (lane la) and one-hot B (lane lb)
// and record which output lanes become 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CBSZ, int ABID, int BLGP>
__global__ void probe(double* out) {
  const int la = blockIdx.x, lb = blockIdx.y, l = threadIdx.x;
  double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, BLGP);
  out[((size_t)la * 64 + lb) * 64 + l] = d;
}
template <int CBSZ, int ABID, int BLGP> void run(double* dout, std::vector<double>& h) {
  hipLaunchKernelGGL((probe<CBSZ, ABID, BLGP>), dim3(64, 64), dim3(64), 0, 0, dout);
  (void)hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  printf("cbsz=%d abid=%d blgp=%d: nonzero (la, lb) -> output lane\n", CBSZ, ABID, BLGP);
  int cnt = 0;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb)
      for (int l = 0; l < 64; ++l)
        if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) {
          if (cnt < 40 || (la % 16 == 5 && cnt < 400)) printf("  a-lane %2d  b-lane %2d -> d-lane %2d\n", la, lb, l);
          ++cnt;
        }
  printf("  total nonzero triples: %d\n", cnt);
}
int main() {
  double* dout; (void)hipMalloc(&dout, 64ull * 64 * 64 * 8);
  std::vector<double> h(64ull * 64 * 64);
  run<0, 0, 0>(dout, h);
  run<2, 0, 0>(dout, h);
  run<2, 1, 0>(dout, h);
  return 0;
}
