"""Actor MLP forward on the MFMA units (nm_policy_forward): the inference half of rsl_rl's ActorCritic
(Linear -> ELU ... -> Linear), torch.nn.Linear weight layout so state_dicts load unchanged."""
import ctypes as C

import torch

from . import _lib


class ActorMLP(torch.nn.Module):
    def __init__(self, dims=(66, 256, 256, 18)):
        super().__init__()
        self.dims = list(dims)
        self.layers = torch.nn.ModuleList(torch.nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1))

    def torch_forward(self, x):
        for i, l in enumerate(self.layers):
            x = l(x)
            if i < len(self.layers) - 1:
                x = torch.nn.functional.elu(x)
        return x

    @torch.no_grad()
    def forward(self, obs):
        L = _lib.load()
        if not obs.is_cuda:
            raise _lib.NightmareHipError("ActorMLP.forward needs a HIP tensor (no CPU path); use torch_forward on the host")
        obs = obs.contiguous().float()
        n = len(self.layers)
        w = (C.c_void_p * n)(*[l.weight.data_ptr() for l in self.layers])
        b = (C.c_void_p * n)(*[l.bias.data_ptr() for l in self.layers])
        dims = (C.c_int32 * (n + 1))(*self.dims)
        out = torch.empty(obs.shape[0], self.dims[-1], device=obs.device, dtype=torch.float32)
        stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
        with torch.cuda.device(obs.device):
            _lib.check(L.nm_policy_forward(obs.data_ptr(), obs.shape[0], w, b, dims, n, out.data_ptr(), stream))
        return out
