"""Time per kernel of a HIP graph of 50 trivial kernels: the launch floor that graph-timed kernels sit on."""
import torch
x = torch.zeros(64, device="cuda")
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): x.add_(1)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(50): x.add_(1)
for _ in range(5): g.replay()
torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
for _ in range(20): g.replay()
e.record(); torch.cuda.synchronize()
print(f"trivial kernel in a graph: {s.elapsed_time(e) / 1000 * 1e3:.2f} us per kernel")
