// Phase timing of the X tiles (V^T as extra block rows of the training factorisation) inside the panel launches, under
// full load: 256 orderings of 1024^2 (512 matrices + 256 X buffers), diagonal launch + 8 panel launches as a C3 step
// runs them; 100 MHz wall-clock stamps at the phase boundaries of 64 workgroups spread over each launch's grid.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ls-spa_amd/csrc -I include -o tools/bin/xtile_probe tools/xtile_probe.hip
#define LSSPA_PANEL_STAMPS 1
#define LSSPA_PST_SLOTS 2048
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

int main(int argc, char** argv) {
  const int flags = argc > 1 ? atoi(argv[1]) : 0;
  const int n_ord = argc > 2 ? atoi(argv[2]) : 256;
  const int p_pad = 1024, n_mats = 2 * n_ord, nblk = p_pad / NB, p_live = 1008;
  double *A, *X, *Dinv, *diag0; int32_t* info;
  (void)hipMalloc(&A, (size_t)n_mats * p_pad * p_pad * 8); (void)hipMalloc(&Dinv, (size_t)n_mats * nblk * 4096 * 8);
  (void)hipMalloc(&X, (size_t)n_ord * p_pad * p_pad * 8);
  (void)hipMalloc(&diag0, (size_t)n_mats * p_pad * 8); (void)hipMalloc(&info, 64); (void)hipMemset(info, 0, 64);
  // the fused lift scan of the X tiles (argv[3] = 0 switches it off): row flags, running sums, partial sums
  const int scan_mode = argc > 3 ? atoi(argv[3]) : 2;
  int32_t* rflags; double *run, *Ppart;
  (void)hipMalloc(&rflags, (size_t)n_mats * 4); (void)hipMalloc(&run, (size_t)n_ord * p_pad * 8);
  (void)hipMalloc(&Ppart, (size_t)n_ord * 16 * p_pad * 8);
  PanelLift pl = {rflags, run, Ppart, (int64_t)16 * p_pad, 1000, scan_mode};
  std::vector<double> d0((size_t)n_mats * p_pad, 2.0);
  (void)hipMemcpy(diag0, d0.data(), d0.size() * 8, hipMemcpyHostToDevice);
  const int n_panel = p_pad / 128 - 1;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(fill_spd, dim3(8192), dim3(256), 0, 0, A, p_pad, n_mats);
    (void)launch_chol2_diag(A, Dinv, diag0, 1e-13, info, p_pad, n_mats, 0, 0, rflags);
    (void)hipDeviceSynchronize();
    for (int Jo = 0; Jo <= n_panel; ++Jo) {
      std::vector<long long> zero(PST_SLOTS * PST_PHASES, 0);
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pstamps), zero.data(), zero.size() * 8);
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0);
      hipError_t le = launch_chol2_panel(A, Dinv, diag0, 1e-13, info, p_pad, Jo, n_mats, 0, 0, flags, p_live, X, n_ord,
                                           scan_mode ? &pl : nullptr);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      if (le != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(le)); return 1; }
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> st(PST_SLOTS * PST_PHASES);
      (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_pstamps), st.size() * 8);
      if (rep < 2) continue;
      const int n_lt = n_panel - Jo, n_x = Jo + 1, lt1 = n_lt > 0 ? n_lt - 1 : 0;
      const int grid = n_mats * n_lt + n_ord * n_x;
      const int stride = grid / PST_SLOTS ? grid / PST_SLOTS : 1;
      printf("Jo=%d  launch %.1f us, grid %d\n", Jo, ms * 1e3, grid);
      long long first = 0;
      for (int s = 0; s < PST_SLOTS; ++s) if (st[s * PST_PHASES] && (!first || st[s * PST_PHASES] < first)) first = st[s * PST_PHASES];
      for (int s = 0; s < PST_SLOTS; ++s) {
        const int id0 = s * stride;
        if (id0 >= grid) break;
        const long long* t = &st[s * PST_PHASES];
        if (t[0] == 0) continue;
        // kind of workgroup id (grouped order)
        const int n0 = n_lt > 0 ? n_mats : 0, n_l = n0 + n_mats * lt1;
        const char* kind = "tile0"; int tile = 0;
        const int raw = id0;
        int id = raw;
        if ((flags & 8) && n0 > 0) {
          const int H = 256, pairs = (n0 + H - 1) / H - 1;
          if (id < pairs * 2 * H) { const int bq = id / (2 * H), r = id - bq * 2 * H; id = r < H ? bq * H + r : n0 + bq * H + (r - H); }
          else { const int r = id - pairs * 2 * H, last = n0 - pairs * H; id = r < last ? pairs * H + r : n0 + pairs * H + (r - last); }
        }
        if (id >= n_l) { kind = "X"; tile = (id - n_l) / n_ord; }
        else if (id >= n0) { kind = "L"; tile = 1 + ((id - n0) % (8 * lt1)) / 8; }
        if (raw % 61) continue;
        printf("   id %5d %-6s tile %d  start %7.1f | init %5.1f  k-loop %6.1f  solve %5.1f  store %5.1f | total %6.1f", raw, kind, tile,
               (t[0] - first) * 0.01, (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01,
               (t[(kind[0] == 't') ? 7 : 6] - t[0]) * 0.01);
        if (kind[0] == 't') printf("  (update %5.1f  factor %5.1f)", (t[6] - t[4]) * 0.01, (t[7] - t[6]) * 0.01);
        if (kind[0] == 'X' && scan_mode) printf("  (waited for row p: %4.2f; store incl. scan)", (t[15] - t[14]) * 0.01);
        printf("\n");
      }
    }
  }
  int32_t h_info = 0; (void)hipMemcpy(&h_info, info, 4, hipMemcpyDeviceToHost);
  printf("info = %d\n", h_info);
  return 0;
}
