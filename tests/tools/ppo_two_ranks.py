"""Data-parallel PPO update on the hand-written kernels, rehearsed with two ranks on ONE GPU (gloo): started by
tests/test_00_bench_multirank.py under torch.distributed.run. Each rank takes half of every mini-batch through
FusedUpdate.minibatch_data_parallel (local gradient, one all-reduce of gradient | KL, the step); afterwards
  * both ranks hold bit-identical parameters, Adam moments and learning rate, and
  * they equal a single-process update on the whole mini-batch (mean of the two half-batch gradients = full-batch gradient)."""
import copy
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nightmare_rl_amd.rl import ActorCritic          # noqa: E402
from nightmare_rl_amd.rl.fused import FusedUpdate    # noqa: E402


def main():
    dist.init_process_group(os.environ.get("NM_DIST_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = "cuda:0"
    torch.manual_seed(11)
    ac = ActorCritic(66, 66, 18, actor_hidden_dims=[54, 42, 30], critic_hidden_dims=[54, 42, 30], activation="elu", init_noise_std=0.8).to(dev)
    ref = copy.deepcopy(ac)
    opt, ropt = torch.optim.Adam(ac.parameters(), lr=1e-3), torch.optim.Adam(ref.parameters(), lr=1e-3)
    fu, rfu = FusedUpdate(ac, opt, dev, lr=1e-3), FusedUpdate(ref, ropt, dev, lr=1e-3)
    hp = dict(clip=0.2, value_coef=1.0, entropy_coef=0.0015, clip_value=True, desired_kl=0.01, adaptive=True, max_grad_norm=1.0)
    B = 4096 + 64                      # rows per rank
    gen = torch.Generator(device=dev).manual_seed(3)
    for it in range(4):
        rn = lambda *s: torch.randn(*s, device=dev, generator=gen)
        obs = rn(world * B, 66)
        with torch.no_grad():
            old_mu = ref.actor(obs) + 0.05 * rn(world * B, 18)
            old_sigma = (ref.std * (1 + 0.05 * rn(18))).expand(world * B, 18).contiguous()
            actions = old_mu + old_sigma * rn(world * B, 18)
            old_logp = torch.distributions.Normal(old_mu, old_sigma).log_prob(actions).sum(-1)
            tv = ref.critic(obs).squeeze(-1) + 0.3 * rn(world * B)
        ret, adv = tv + rn(world * B), rn(world * B)
        full = (obs, actions, tv, adv, ret, old_logp, old_mu, old_sigma)
        mine = tuple(t[rank * B:(rank + 1) * B].contiguous() for t in full)
        fu.minibatch_data_parallel(*mine, hp=hp, world=world)
        rfu.minibatch(*full, hp)
        st, rst = fu.read_state(), rfu.read_state()
        assert abs(st["kl"] - rst["kl"]) < 1e-6 + 1e-4 * rst["kl"], (st["kl"], rst["kl"])
        assert st["lr"] == rst["lr"] and abs(st["grad_norm"] - rst["grad_norm"]) < 1e-4 * rst["grad_norm"], (st, rst)
        torch.testing.assert_close(fu.flat, rfu.flat, atol=2e-6, rtol=1e-5)
    both = [torch.empty_like(fu.flat) for _ in range(world)]
    dist.all_gather(both, fu.flat)
    assert all(torch.equal(both[0], b) for b in both[1:]), "ranks diverged"
    mom = [torch.empty_like(fu.m) for _ in range(world)]
    dist.all_gather(mom, fu.m)
    assert all(torch.equal(mom[0], b) for b in mom[1:])
    dist.barrier()
    if rank == 0:
        print("PPO_TWO_RANKS_OK lr=%g kl=%g" % (st["lr"], st["kl"]), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
