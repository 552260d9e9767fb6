"""Actor MLP forward on the MFMA units (nm_policy_forward): the inference half of rsl_rl's ActorCritic
(Linear -> ELU ... -> Linear), torch.nn.Linear weight layout so state_dicts load unchanged."""
import ctypes as C

import torch

from . import _lib


class ActorMLP(torch.nn.Module):
    _packed = None   # (id of the network whose weights sit in the library's packed buffer, its version key)

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

    def _bind(self, n, device):
        """(Re)build the cached C argument arrays: weight pointers, dims, and a persistent output buffer (stable
        addresses, so a policy+env step can be captured in a HIP graph)."""
        ptrs = tuple(p.data_ptr() for l in self.layers for p in (l.weight, l.bias))
        key = (n, str(device), ptrs)
        if getattr(self, "_key", None) != key:
            k = len(self.layers)
            self._w = (C.c_void_p * k)(*[l.weight.data_ptr() for l in self.layers])
            self._b = (C.c_void_p * k)(*[l.bias.data_ptr() for l in self.layers])
            self._dims = (C.c_int32 * (k + 1))(*self.dims)
            self._out = torch.empty(n, self.dims[-1], device=device, dtype=torch.float32)
            self._key = key

    @torch.no_grad()
    def forward(self, obs):
        L = _lib.load()
        if not obs.is_cuda:
            raise _lib.NightmareHipError("ActorMLP.forward needs a HIP tensor (no CPU path); use torch_forward on the host")
        if obs.dtype != torch.float32 or not obs.is_contiguous():
            obs = obs.contiguous().float()
        self._bind(obs.shape[0], obs.device)
        stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
        if max(self.dims) > 256 or len(self.layers) > 4:     # per-layer kernels
            _lib.check(L.nm_policy_forward(obs.data_ptr(), obs.shape[0], self._w, self._b, self._dims, len(self.layers),
                                           self._out.data_ptr(), stream))
            return self._out
        # weights are repacked for the fused kernel only when they changed (in-place updates bump tensor versions) or when
        # another network was packed in between: a rollout pays one launch per policy step
        ver = (self._key, tuple(p._version for l in self.layers for p in (l.weight, l.bias)))
        if ActorMLP._packed != (id(self), ver):
            _lib.check(L.nm_policy_pack(self._w, self._b, self._dims, len(self.layers), stream))
            ActorMLP._packed = (id(self), ver)
        _lib.check(L.nm_policy_forward_packed(obs.data_ptr(), obs.shape[0], self._out.data_ptr(), stream))
        return self._out
