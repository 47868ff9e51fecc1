"""The Gram reduction at any shape: python3 tools/gram_variants.py [p rows | c5]  (developer tool)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ls-spa_amd"))
import torch, numpy as np
from ls_spa._engine import HipEngine
if len(sys.argv) > 2:
    p, N = int(sys.argv[1]), int(sys.argv[2])
else:
    p, N = (5000, 200000) if (len(sys.argv) > 1 and sys.argv[1] == "c5") else (1000, 100000)
f32 = p == 5000
dev = torch.device("cuda:0"); torch.manual_seed(0)
dt = torch.float32 if f32 else torch.float64
Xa = torch.randn(N, p, dtype=dt, device=dev); ya = torch.randn(N, dtype=dt, device=dev)
torch.cuda.synchronize()
eng = HipEngine(0); eng.profile(True)
for rnd in range(6):
    eng.profile_reset()
    eng.load_device_data(Xa.data_ptr(), p, ya.data_ptr(), N, Xa.data_ptr(), p, ya.data_ptr(), N, p, 0.0, f32=f32)
    eng.synchronize()
ms, cnt = eng.profile_read()["gram"]
print(p, N, "gram ms per side", ms / cnt,
      "TFLOP/s", N * (p + 1) * (p + 2) / (ms / cnt * 1e-3) / 1e12)
