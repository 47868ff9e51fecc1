"""Time the Gram reduction at the C3 shape (developer tool)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ls-spa_amd"))
import torch, numpy as np
from ls_spa._engine import HipEngine
p, N = 1000, 100000
dev = torch.device("cuda:0"); torch.manual_seed(0)
Xa = torch.randn(N, p, dtype=torch.float64, device=dev); Xe = torch.randn(N, p, dtype=torch.float64, device=dev)
ya = torch.randn(N, dtype=torch.float64, device=dev); ye = torch.randn(N, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
eng = HipEngine(0); eng.profile(True)
for flags in (0, 65536, 0, 65536):
    eng.set_flags(flags)
    for rnd in range(3):
        eng.profile_reset()
        eng.load_device_data(Xa.data_ptr(), p, ya.data_ptr(), N, Xe.data_ptr(), p, ye.data_ptr(), N, p, 0.0)
        eng.synchronize()
    print("flags", flags, "gram (ms, launches)", eng.profile_read()["gram"])
eng.set_flags(0)
for rnd in range(2):
    eng.profile_reset()
    eng.load_device_data(Xa.data_ptr(), p, ya.data_ptr(), N, Xe.data_ptr(), p, ye.data_ptr(), N, p, 0.0)
    eng.synchronize()
    print("gram (ms, launches)", eng.profile_read()["gram"])
G = (Xa.T @ Xa / N).cpu().numpy(); print("max err", np.abs(eng.gram()[0] - G).max())

if len(sys.argv) > 1 and sys.argv[1] == "c3":
    sys.exit(0)
# the C5 shape in float32 (p = 5000, 200000 rows)
del Xa, Xe, ya, ye
torch.cuda.empty_cache()
p, N = 5000, 200000
Xa = torch.randn(N, p, dtype=torch.float32, device=dev); ya = torch.randn(N, dtype=torch.float32, device=dev)
torch.cuda.synchronize()
for rnd, flags in enumerate((0, 65536, 0, 65536)):
    eng.set_flags(flags)
    eng.profile_reset()
    eng.load_device_data(Xa.data_ptr(), p, ya.data_ptr(), N, Xa.data_ptr(), p, ya.data_ptr(), N, p, 0.01, f32=True)
    eng.synchronize()
    ms, cnt = eng.profile_read()["gram"]
    print("C5 flags", flags, "gram (ms, launches)", ms, cnt, "TFLOP/s per side", N * (p + 1) * (p + 2) / (ms / cnt * 1e-3) / 1e12)
