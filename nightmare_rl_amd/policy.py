"""MLP forward on the MFMA units (nm_policy_*): the inference half of rsl_rl's ActorCritic (Linear -> ELU ... -> Linear),
torch.nn.Linear weight layout so state_dicts load unchanged.

Each network owns ONE library handle with a packed copy of its parameters. The copy is refreshed by `repack()`: explicitly (after
an optimiser step, `load_state_dict`, or any write through `.data`) or, in `forward`, whenever the parameters' generation changed -
detected by a cheap fingerprint (data pointers + tensor versions) plus `mark_dirty()` for writes the versions do not see. A HIP graph
that captures `forward` replays the kernel only; after changing weights call `repack()` before the next replay."""
import ctypes as C

import torch

from . import _lib


class PackedMLP:
    """Handle of one packed network on one device (no torch module semantics): load(weights, biases) then forward(obs)."""

    def __init__(self, dims, device):
        self.dims = [int(d) for d in dims]
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.NightmareHipError("the MLP kernels need a HIP device (no CPU path)")
        self._L = _lib.load()
        h = C.c_void_p()
        arr = (C.c_int32 * len(self.dims))(*self.dims)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self._L.nm_policy_create(arr, len(self.dims) - 1, idx, C.byref(h)))
        self._h = h
        self._out = None

    def load(self, weights, biases):
        k = len(self.dims) - 1
        assert len(weights) == len(biases) == k
        ws = [w.detach().to(device=self.device, dtype=torch.float32).contiguous() for w in weights]
        bs = [b.detach().to(device=self.device, dtype=torch.float32).contiguous() for b in biases]
        for l in range(k):
            assert tuple(ws[l].shape) == (self.dims[l + 1], self.dims[l]) and tuple(bs[l].shape) == (self.dims[l + 1],)
        wp = (C.c_void_p * k)(*[w.data_ptr() for w in ws])
        bp = (C.c_void_p * k)(*[b.data_ptr() for b in bs])
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._L.nm_policy_load(self._h, wp, bp, stream))
        self._keep = (ws, bs)     # alive until the stream-ordered copy has run

    def forward(self, obs, out=None):
        if obs.dtype != torch.float32 or not obs.is_contiguous():
            obs = obs.contiguous().float()
        n = obs.shape[0]
        if out is None:
            if self._out is None or self._out.shape[0] != n:
                self._out = torch.empty(n, self.dims[-1], device=self.device, dtype=torch.float32)   # stable address: graph capture
            out = self._out
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._L.nm_policy_forward(self._h, obs.data_ptr(), n, out.data_ptr(), stream))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._L.nm_policy_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ActorMLP(torch.nn.Module):
    def __init__(self, dims=(66, 256, 256, 18)):
        super().__init__()
        self.dims = list(dims)
        self.layers = torch.nn.ModuleList(torch.nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1))
        self._packed = None
        self._generation = None
        self._dirty = True

    def torch_forward(self, x):
        for i, l in enumerate(self.layers):
            x = l(x)
            if i < len(self.layers) - 1:
                x = torch.nn.functional.elu(x)
        return x

    def _fingerprint(self):
        return tuple((p.data_ptr(), p._version) for l in self.layers for p in (l.weight, l.bias))

    def mark_dirty(self):
        """Tell the module its parameters changed in a way tensor versions do not record (writes through `.data`, `dist.broadcast`)."""
        self._dirty = True

    def repack(self):
        dev = self.layers[0].weight.device
        if self._packed is None or self._packed.device != dev:
            self._packed = PackedMLP(self.dims, dev)
        self._packed.load([l.weight for l in self.layers], [l.bias for l in self.layers])
        self._generation = self._fingerprint()
        self._dirty = False

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._dirty = True

    @torch.no_grad()
    def forward(self, obs):
        if not obs.is_cuda:
            raise _lib.NightmareHipError("ActorMLP.forward needs a HIP tensor (no CPU path); use torch_forward on the host")
        if self._dirty or self._packed is None or self._generation != self._fingerprint() or self._packed.device != obs.device:
            self.repack()
        return self._packed.forward(obs)
