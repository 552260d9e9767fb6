"""Kernel time of the fused MLP forward: 50 forwards captured in one HIP graph (no host launch path), HIP-event timed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.policy import ActorMLP
torch.manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NETS = ([66, 256, 256, 18], [66, 54, 42, 30, 18], [66, 108, 84, 60, 19])
if len(sys.argv) > 2:       # "big": BASELINE config 3's network only (one kernel shape per counter run)
    NETS = NETS[:1] if sys.argv[2] == "big" else NETS[1:]
for dims in NETS:
    net = ActorMLP(dims).cuda(); x = torch.randn(N, 66, device="cuda")
    with torch.no_grad():
        ref = net.torch_forward(x)
    out = net(x)
    err = float((out - ref).abs().max())
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): net(x)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): net(x)
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(20): g.replay()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 1000 * 1e3
    fl = 2 * N * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    print(f"{dims}: max err {err:.1e}  {us:7.2f} us/forward  {fl / us / 1e6:7.2f} TFLOP/s = {100 * fl / us / 1e6 / 157.3:.1f} % of the 157.3 TF f32-MFMA peak")
