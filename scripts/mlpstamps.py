"""Cycle stamps of workgroup 0 of the fused MLP kernel (measurement build: make variant NAME=mlpst EXTRA=-DNM_MLP_STAMPS;
NM_HIP_LIB=nightmare_rl_amd/csrc/libnightmare_hip_mlpst.so python scripts/mlpstamps.py)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.policy import ActorMLP
from nightmare_rl_amd import _lib
L = _lib.load()
for dims in ([66, 256, 256, 18], [66, 54, 42, 30, 18]):
    net = ActorMLP(dims).cuda(); x = torch.randn(4096, 66, device="cuda")
    for _ in range(5): net(x)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    L.nm_mlp_read_stamps(out)
    t = list(out)
    names = ["start", "obs staged"] + [f"L{l} {w}" for l in range(len(dims) - 1) for w in ("mfma done", "reduced", "epilogue+sync")]
    print(dims)
    for i in range(1, 2 + 3 * (len(dims) - 1)):
        print(f"   {names[i]:18s} +{t[i] - t[i - 1]:6d} cycles   (t = {t[i] - t[0]})")
