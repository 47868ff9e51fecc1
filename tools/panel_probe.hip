// Phase timing of chol_panel2 under full load (developer tool): the whole factorisation of 512 matrices of 1024^2
// (diagonal launch + 7 panel launches, as a C3 step runs them), with 100 MHz wall-clock stamps at the phase
// boundaries of 64 workgroups spread over each launch's grid.  Prints, per launch, the median duration of each phase
// over the sampled workgroups (tile-0 workgroups, which also factor the next diagonal block, apart).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ls-spa_amd/csrc -I include -o tools/bin/panel_probe tools/panel_probe.hip
#define LSSPA_PANEL_STAMPS 1
#include "../ls-spa_amd/csrc/k_factor.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
using namespace lsspa;

__global__ void fill_spd(double* A, int p_pad, int n_mats) {
  const size_t total = (size_t)n_mats * p_pad * p_pad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i % ((size_t)p_pad * p_pad);
    const int chunk = (int)(e / ((size_t)p_pad * 16)), r = (int)((e / 16) % p_pad), c = chunk * 16 + (int)(e % 16);
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull;
    z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
    const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    A[i] = (r == c) ? 2.0 : (c < r ? 1e-3 * u : 0.0);
  }
}

int main() {
  const int p_pad = 1024, n_mats = 512, nblk = p_pad / NB, p_live = 1008;
  double *A, *Dinv, *diag0; int32_t* info;
  (void)hipMalloc(&A, (size_t)n_mats * p_pad * p_pad * 8); (void)hipMalloc(&Dinv, (size_t)n_mats * nblk * 4096 * 8);
  (void)hipMalloc(&diag0, (size_t)n_mats * p_pad * 8); (void)hipMalloc(&info, 64); (void)hipMemset(info, 0, 64);
  std::vector<double> d0((size_t)n_mats * p_pad, 2.0);
  (void)hipMemcpy(diag0, d0.data(), d0.size() * 8, hipMemcpyHostToDevice);
  const char* names[7] = {"init (A tile -> acc)", "k-loop", "solve", "store loop", "diag update (restage)", "diag update (store)",
                          "diag factorisation"};
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(fill_spd, dim3(8192), dim3(256), 0, 0, A, p_pad, n_mats);
    (void)launch_chol2_diag(A, Dinv, diag0, 1e-13, info, p_pad, n_mats, 0, 0);
    (void)hipDeviceSynchronize();
    for (int Jo = 0; Jo < p_pad / 128 - 1; ++Jo) {
      std::vector<long long> zero(PST_SLOTS * PST_PHASES, 0);
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pstamps), zero.data(), zero.size() * 8);
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0);
      (void)launch_chol2_panel(A, Dinv, diag0, 1e-13, info, p_pad, Jo, n_mats, 0, 0, 0, p_live);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> st(PST_SLOTS * PST_PHASES);
      (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_pstamps), st.size() * 8);
      if (rep < 2) continue;
      const int n_tiles = p_pad / 128 - 1 - Jo, grid = n_mats * n_tiles;
      const int stride = grid / PST_SLOTS ? grid / PST_SLOTS : 1;
      std::vector<double> ph[2][8];
      long long first = 0;
      for (int s = 0; s < PST_SLOTS; ++s) {
        const int id = s * stride;
        if (id >= grid) break;
        const long long* t = &st[s * PST_PHASES];
        if (t[0] == 0) continue;
        if (first == 0 || t[0] < first) first = t[0];
        const int is0 = id < n_mats ? 1 : 0;
        for (int k = 0; k < 6; ++k) ph[is0][k].push_back((t[k + 1] - t[k]) * 0.01);
        if (is0) ph[1][6].push_back((t[7] - t[6]) * 0.01);
        ph[is0][7].push_back((t[is0 ? 7 : 6] - t[0]) * 0.01);
      }
      auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
      printf("Jo=%d  launch %.1f us, %d tiles x %d matrices; per phase, median us over the sampled workgroups [other tiles | tile 0]\n",
             Jo, ms * 1e3, n_tiles, n_mats);
      for (int k = 0; k < 7; ++k)
        printf("    %-24s %7.1f | %7.1f\n", names[k], k < 6 ? med(ph[0][k]) : 0.0, med(ph[1][k]));
      {   // finer stamps: solve = [first block in LDS | tri_mult 0 | full product | tri_mult 1]; diag update = [fence + initial
          // loads issued + barrier | first chunk landed and staged | 2nd | 3rd iteration | the other five + tail]
        std::vector<double> f[9];
        for (int s2 = 0; s2 < PST_SLOTS; ++s2) {
          const int id = s2 * stride;
          if (id >= grid || id < n_mats) continue;
          const long long* t = &st[s2 * PST_PHASES];
          if (t[0] == 0) continue;
          f[0].push_back((t[12] - t[2]) * 0.01); f[1].push_back((t[13] - t[12]) * 0.01); f[2].push_back((t[14] - t[13]) * 0.01);
          f[3].push_back((t[3] - t[14]) * 0.01);
          f[4].push_back((t[8] - t[4]) * 0.01); f[5].push_back((t[9] - t[8]) * 0.01); f[6].push_back((t[10] - t[9]) * 0.01);
          f[7].push_back((t[11] - t[10]) * 0.01); f[8].push_back((t[5] - t[11]) * 0.01);
        }
        printf("      solve: %.1f | %.1f | %.1f | %.1f      diag update: %.1f | %.1f | %.1f | %.1f | %.1f\n", med(f[0]), med(f[1]), med(f[2]),
               med(f[3]), med(f[4]), med(f[5]), med(f[6]), med(f[7]), med(f[8]));
      }
      printf("    %-24s %7.1f | %7.1f   (%zu | %zu workgroups sampled)\n", "whole workgroup", med(ph[0][7]), med(ph[1][7]),
             ph[0][7].size(), ph[1][7].size());
    }
  }
  int32_t h_info = 0; (void)hipMemcpy(&h_info, info, 4, hipMemcpyDeviceToHost);
  printf("info = %d\n", h_info);
  return 0;
}
