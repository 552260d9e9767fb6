"""Where a wave of the PPO fwd/bwd kernel spends its time (s_memtime stamps; needs the -DNM_PPO_STAMPS measurement build:
   make -C nightmare_rl_amd/csrc variant NAME=ppostamps EXTRA=-DNM_PPO_STAMPS ; NM_HIP_LIB=.../libnightmare_hip_ppostamps.so python scripts/ppostamps.py)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd import _lib
from nightmare_rl_amd.rl import ActorCritic
from nightmare_rl_amd.rl.fused import FusedUpdate
torch.manual_seed(0)
B = int(os.environ.get("PPO_B", "81920"))
ac = ActorCritic(66, 66, 18, actor_hidden_dims=[54, 42, 30], critic_hidden_dims=[54, 42, 30], activation="elu", init_noise_std=1.0).cuda()
opt = torch.optim.Adam(ac.parameters(), lr=1e-3)
fu = FusedUpdate(ac, opt, "cuda:0", lr=1e-3)
hp = dict(clip=0.2, value_coef=1.0, entropy_coef=0.0, clip_value=True, desired_kl=0.01, adaptive=True, max_grad_norm=1.0)
obs = torch.randn(B, 66, device="cuda"); mu = torch.randn(B, 18, device="cuda") * 0.1; sg = torch.ones(B, 18, device="cuda")
act = mu + torch.randn(B, 18, device="cuda"); lp = torch.distributions.Normal(mu, sg).log_prob(act).sum(-1)
tv = torch.randn(B, device="cuda"); adv = torch.randn(B, device="cuda"); ret = tv + torch.randn(B, device="cuda")
L = _lib.load()
for _ in range(3): fu.minibatch(obs, act, tv, adv, ret, lp, mu, sg, hp, phase=1)
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
if hasattr(L, "nm_ppo_read_stamps") or True:
    try:
        f = L.nm_ppo_read_stamps
        f(out, 1)
    except AttributeError:
        f = None
K = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K): fu.minibatch(obs, act, tv, adv, ret, lp, mu, sg, hp, phase=1)
e1.record(); torch.cuda.synchronize()
print(f"fwdbwd + reduce: {e0.elapsed_time(e1) / K * 1000:.1f} us per mini-batch of {B}")
e0.record()
for _ in range(K): fu.minibatch(obs, act, tv, adv, ret, lp, mu, sg, hp, phase=0)
e1.record(); torch.cuda.synchronize()
print(f"fwdbwd + step (the whole mini-batch): {e0.elapsed_time(e1) / K * 1000:.1f} us per mini-batch of {B}")
if f:
    f(out, 0)
    names = ["loads", "fwd L0", "fwd L1", "fwd L2", "fwd L3", "head", "bwd L3 park+dX", "bwd L3 barrier+dW", "bwd L2 park+dX", "bwd L2 barrier+dW",
             "bwd L1 park+dX", "bwd L1 barrier+dW", "bwd L0 park", "bwd L0 barrier+dW", "epilogue", "loop"]
    order = [15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14]
    tot = sum(out)
    nm = ["loads", "fwd L0", "fwd L1", "fwd L2", "fwd L3", "head", "bwd L3 park+dX", "bwd L3 barrier+dW", "bwd L2 park+dX", "bwd L2 barrier+dW", "bwd L1 park+dX",
          "bwd L1 barrier+dW", "bwd L0 park", "bwd L0 barrier+dW (to 13)", "epilogue", "pass loop top"]
    for k in range(16):
        print(f"{k:2d} {nm[k]:28s} {out[k] / K / 512:10.0f} ticks per workgroup-launch (wave 0 of the 512 workgroups: actor and critic blocks together) {100.0 * out[k] / tot:5.1f} %")
