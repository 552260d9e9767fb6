import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.policy import ActorMLP
torch.manual_seed(0)
for dims in ([66, 256, 256, 18], [66, 54, 42, 30, 18]):
    net = ActorMLP(dims).cuda(); x = torch.randn(4096, 66, device="cuda")
    ref = net.torch_forward(x); out = net(x)
    print(dims, "max err", float((out - ref).abs().max()))
    for f, name in ((net, "mfma fused"), (net.torch_forward, "torch")):
        for _ in range(20): f(x)
        torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
        for _ in range(200): f(x)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 200 * 1e3
        fl = 2 * 4096 * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
        print(f"   {name:12s} {us:8.1f} us/forward  {fl / us / 1e6:8.2f} TFLOP/s")
