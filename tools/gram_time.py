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
for rnd in range(4):
    eng.profile_reset()
    eng.load_device_data(Xa.data_ptr(), p, ya.data_ptr(), N, Xe.data_ptr(), p, ye.data_ptr(), N, p, 0.0)
    eng.synchronize()
    print("gram (ms, launches)", eng.profile_read()["gram"])
G = (Xa.T @ Xa / N).cpu().numpy(); print("max err", np.abs(eng.gram()[0] - G).max())
