"""Per-step cycle stamps of every wave of workgroup 0 in the fused MLP kernel (measurement build -DNM_MLP_STAMPS, see mlpstamps.py)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.policy import ActorMLP
from nightmare_rl_amd import _lib
L = _lib.load()
net = ActorMLP([66, 256, 256, 18]).cuda(); x = torch.randn(int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 66, device="cuda")
for _ in range(5): net(x)
torch.cuda.synchronize()
out = (C.c_ulonglong * (4 * 8 * 20))()
L.nm_mlp_read_steps(out)
st = (C.c_ulonglong * 16)(); L.nm_mlp_read_stamps(st)
t0 = st[0]
import numpy as np
a = np.array(list(out), dtype=np.int64).reshape(4, 8, 20)
for l in range(3):
    print(f"layer {l}: per wave: layer top, plan done, step starts (relative to kernel start), end of steps")
    for w in range(8):
        r = a[l, w]
        steps = [int(r[k] - t0) for k in range(16) if r[k] > 0]
        print(f"  wave {w}: top {int(r[18]-t0):6d} plan {int(r[19]-t0):6d} | " + " ".join(f"{s:6d}" for s in steps) + f" | end {int(r[16]-t0):6d}")
