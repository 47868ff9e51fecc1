// Phase timing of the fused small-p kernel (workgroup 0's wall-clock stamps) on a synthetic SPD problem.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLSSPA_SMALL_STAMPS -I ls-spa_amd/csrc tools/small_probe.hip -o tools/bin/small_probe
#include "../ls-spa_amd/csrc/k_small.hip"
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
#include <numeric>
using namespace lsspa;
int main(int argc, char** argv) {
  const int p = argc > 1 ? atoi(argv[1]) : 100, n_ord = argc > 2 ? atoi(argv[2]) : 256;
  const int ld = ((p + 1 + 127) / 128) * 128;
  std::mt19937_64 rng(1);
  std::normal_distribution<double> nd;
  std::vector<double> G((size_t)p * ld, 0.0), g(ld, 0.0);
  for (int i = 0; i < p; ++i) {
    for (int j = 0; j <= i; ++j) { double v = (i == j) ? 1.0 + 0.01 * nd(rng) : 0.003 * nd(rng); G[(size_t)i * ld + j] = v; G[(size_t)j * ld + i] = v; }
    g[i] = 0.1 * nd(rng);
  }
  std::vector<int32_t> perms((size_t)n_ord * p);
  for (int o = 0; o < n_ord; ++o) { auto b = perms.begin() + (size_t)o * p; std::iota(b, b + p, 0); std::shuffle(b, b + p, rng); }
  double *dG, *dg, *dl; int32_t *dp, *di;
  hipMalloc(&dG, G.size() * 8); hipMalloc(&dg, g.size() * 8); hipMalloc(&dl, (size_t)n_ord * p * 8); hipMalloc(&dp, perms.size() * 4); hipMalloc(&di, 32);
  hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dg, g.data(), g.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dp, perms.data(), perms.size() * 4, hipMemcpyHostToDevice); hipMemset(di, 0, 32); hipMemset(dl, 0, (size_t)n_ord * p * 8);
  SmallArgs a; a.S[0] = a.S[1] = dG; a.s[0] = a.s[1] = dg; a.aug[0] = a.aug[1] = 10.0; a.ld_src = ld; a.perms = dp; a.p = p; a.nb = (p + 16) / 16;
  a.variant = argc > 3 ? atoi(argv[3]) : 0; a.fwd_only = 0; a.r2 = 0.0; a.sum_tol = -1.0; a.sum_quiet = 0.0;
  a.n_ord = n_ord; a.per_sample = 1; a.lifts = dl; a.y_norm_sq = 5.0; a.piv_tol = 1e-12; a.info = di;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); hipError_t e = launch_small_p(a, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long st[16]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_small_stamps), sizeof st);
    printf("launch %s  %.1f us total | perm %.1f gather %.1f chol %.1f zcopy %.1f vsolve %.1f terms %.1f sums %.1f (us, 100 MHz clock)\n", hipGetErrorString(e), ms * 1e3,
           (st[1]-st[0])/100.0, (st[2]-st[1])/100.0, (st[3]-st[2])/100.0, (st[4]-st[3])/100.0, (st[5]-st[4])/100.0, (st[6]-st[5])/100.0, (st[7]-st[6])/100.0);
  }
  {
    long long st[16]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_small_stamps), sizeof st);
    long long cy[16]; hipMemcpyFromSymbol(cy, HIP_SYMBOL(g_small_cycles), sizeof cy);
    printf("shader clock over the kernel: %.3f GHz; first factor16: %lld cycles, second: %lld cycles\n",
           (cy[7] - cy[0]) / ((st[7] - st[0]) * 10.0), cy[8] - cy[2], cy[11] - cy[10]);
    long long hh[16]; hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_small_helper), sizeof hh);
    printf("helper wave 1, block step 0 (cycles): first 4 tiles %lld, next %lld, rest %lld\n", hh[1] - hh[0], hh[2] - hh[1], hh[3] - hh[2]);
    long long ff[4]; hipMemcpyFromSymbol(ff, HIP_SYMBOL(g_small_fstamp), sizeof ff);
    printf("inside the last 16 x 16 factorisation of workgroup 0: the pivot sweep alone %lld cycles\n", ff[1] - ff[0]);
    printf("cholesky detail (us): first factor16 %.2f | barrier+panel(0) %.2f | own trailing tile %.2f | factor16 %.2f | wait for helpers %.2f\n",
           (st[8]-st[2])/100.0, (st[9]-st[8])/100.0, (st[10]-st[9])/100.0, (st[11]-st[10])/100.0, (st[12]-st[11])/100.0);
  }
  if (a.variant == 0 && a.nb <= 7) {
    long long rs[2][12]; hipMemcpyFromSymbol(rs, HIP_SYMBOL(g_reg_stamps), sizeof rs);
    for (int w = 0; w < 2; ++w)
      printf("   gather detail, wave %d: perm + rhs %lld | indices, first fetch issued %lld | block columns %lld | edge %lld\n", w, rs[w][8] - rs[w][0], rs[w][9] - rs[w][8], rs[w][10] - rs[w][9], rs[w][1] - rs[w][10]);
    long long cs[2][8][4]; hipMemcpyFromSymbol(cs, HIP_SYMBOL(g_reg_chol), sizeof cs);
    for (int kb = 0; kb + 1 < a.nb; ++kb)
      printf("   block step %d, wave 0: elimination %lld | scaling, inverse to operand form %lld | panel + trailing products %lld\n", kb,
             cs[0][kb][1] - cs[0][kb][0], cs[0][kb][2] - cs[0][kb][1], cs[0][kb + 1][0] - cs[0][kb][2]);
    for (int w = 0; w < 2; ++w)
      printf("register kernel, wave %d (cycles): gather %lld | cholesky %lld | hand-over %lld | %s %lld | tail %lld | total %lld\n", w,
             rs[w][1] - rs[w][0], rs[w][2] - rs[w][1], rs[w][3] - rs[w][2], w ? "lift scan behind the solve" : "V solve", rs[w][4] - rs[w][3],
             rs[w][7] - rs[w][4], rs[w][7] - rs[w][0]);
  }
  std::vector<double> hl((size_t)n_ord * p); hipMemcpy(hl.data(), dl, hl.size() * 8, hipMemcpyDeviceToHost);
  double cs = 0; for (double v : hl) cs += v; printf("checksum of the lifts %.15e\n", cs);
  int info; hipMemcpy(&info, di, 4, hipMemcpyDeviceToHost); printf("info %d lds %zu\n", info, small_p_lds_bytes(a.nb));
  return 0;
}
