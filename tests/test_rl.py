"""The on-policy stack (rsl_rl-shaped) on the host: GAE against the textbook recursion, checkpoint key layout,
and that PPO actually learns on a toy vectorised env with the NightmareV3 PPO hyper-parameters."""
import os

import numpy as np
import pytest
import torch

from nightmare_rl_amd.envs.helpers import class_to_dict
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3ConfigPPO
from nightmare_rl_amd.rl import ActorCritic, OnPolicyRunner, RolloutStorage


class ToyEnv:
    """obs = target (3 numbers); reward = -|action - target|^2; episodes of 20 steps. Same surface as NightmareV3Env."""

    def __init__(self, n):
        self.num_envs, self.num_obs, self.num_privileged_obs, self.num_actions = n, 3, None, 3
        self.max_episode_length = 20
        self.episode_length_buf = torch.zeros(n, dtype=torch.int64)
        self.t = torch.rand(n, 3) * 2 - 1
        self.extras = {}

    def get_observations(self):
        return self.t.clone()

    def get_privileged_observations(self):
        return None

    def reset(self):
        return self.get_observations(), None

    def step(self, a):
        rew = -((a - self.t) ** 2).sum(dim=1)
        self.episode_length_buf += 1
        to = self.episode_length_buf > self.max_episode_length
        done = to.long()
        ids = done.nonzero()[:, 0]
        self.t[ids] = torch.rand(len(ids), 3) * 2 - 1
        self.episode_length_buf[ids] = 0
        self.extras["time_outs"] = to.float()
        return self.get_observations(), None, rew, done, self.extras


def test_gae_matches_textbook_recursion():
    T, N = 7, 5
    g = torch.Generator().manual_seed(0)
    st = RolloutStorage(N, T, [3], [None], [2])
    st.rewards.copy_(torch.randn(T, N, 1, generator=g))
    st.values.copy_(torch.randn(T, N, 1, generator=g))
    st.dones.copy_((torch.rand(T, N, 1, generator=g) < 0.2).to(torch.uint8))
    last = torch.randn(N, 1, generator=g)
    st.compute_returns(last, 0.99, 0.95)
    ret = np.zeros((T, N))
    for e in range(N):
        adv, nxt = 0.0, float(last[e])
        for s in reversed(range(T)):
            live = 1.0 - float(st.dones[s, e])
            delta = float(st.rewards[s, e]) + live * 0.99 * nxt - float(st.values[s, e])
            adv = delta + live * 0.99 * 0.95 * adv
            ret[s, e] = adv + float(st.values[s, e])
            nxt = float(st.values[s, e])
    np.testing.assert_allclose(st.returns[..., 0].numpy(), ret, atol=1e-5)
    a = st.advantages
    assert abs(float(a.mean())) < 1e-5 and abs(float(a.std()) - 1) < 1e-3


def test_checkpoint_layout_matches_rsl_rl(tmp_path):
    ac = ActorCritic(66, 66, 18, actor_hidden_dims=[54, 42, 30], critic_hidden_dims=[54, 42, 30], activation="elu", init_noise_std=1.0)
    keys = list(ac.state_dict())
    assert keys[0] == "std" and "actor.0.weight" in keys and "actor.6.bias" in keys and "critic.6.weight" in keys
    assert ac.state_dict()["actor.0.weight"].shape == (54, 66) and ac.state_dict()["critic.6.weight"].shape == (1, 30)
    assert sum(p.numel() for p in ac.parameters()) == 15043        # 7 776 + 7 249 + 18


def test_ppo_learns_toy_task(tmp_path):
    torch.manual_seed(0)
    cfg = class_to_dict(NightmareV3ConfigPPO())
    cfg["runner"]["num_steps_per_env"] = 24
    cfg["runner"]["save_interval"] = 1000
    env = ToyEnv(256)
    runner = OnPolicyRunner(env, cfg, log_dir=str(tmp_path), device="cpu")
    runner.learn(40, init_at_random_ep_len=True)
    h = runner.history
    assert np.isfinite([r["value_loss"] for r in h]).all()
    first = np.mean([r["mean_reward"] for r in h[2:6]])
    last = np.mean([r["mean_reward"] for r in h[-4:]])
    assert last > first + 5, (first, last)                          # returns improve by a wide margin
    # save / load round trip in the reference's checkpoint format
    path = os.path.join(str(tmp_path), "model_40.pt")
    assert os.path.exists(path) and os.path.exists(os.path.join(str(tmp_path), "progress.jsonl"))
    d = torch.load(path)
    assert set(d) == {"model_state_dict", "optimizer_state_dict", "iter", "infos"} and d["iter"] == 40
    r2 = OnPolicyRunner(ToyEnv(8), cfg, log_dir=None, device="cpu")
    r2.load(path)
    pol = r2.get_inference_policy()
    x = torch.rand(8, 3)
    torch.testing.assert_close(pol(x), runner.alg.actor_critic.act_inference(x))


class BufferReusingEnv(ToyEnv):
    """Like the HIP env: observations are views of ONE persistent buffer that the next step() overwrites."""

    def __init__(self, n):
        super().__init__(n)
        self.obs_buf = self.t.clone()

    def get_observations(self):
        self.obs_buf.copy_(self.t)
        return self.obs_buf

    def step(self, a):
        self.t += 0.01                       # the observation changes every step
        return super().step(a)


def test_rollout_keeps_the_observation_the_policy_saw():
    """Regression: with an env that reuses its observation buffer, the stored (obs, mu, sigma) must stay consistent, otherwise the
    KL estimate of the adaptive learning-rate schedule is garbage and the rate collapses to its floor."""
    torch.manual_seed(0)
    cfg = class_to_dict(NightmareV3ConfigPPO())
    cfg["runner"]["num_steps_per_env"] = 8
    env = BufferReusingEnv(64)
    runner = OnPolicyRunner(env, cfg, log_dir=None, device="cpu")
    alg = runner.alg
    obs = env.get_observations()
    seen = []
    with torch.inference_mode():
        for _ in range(8):
            seen.append(obs.clone())
            a = alg.act(obs, obs)
            obs, _, rew, done, infos = env.step(a)
            alg.process_env_step(rew, done, infos)
    stored = alg.storage.observations
    torch.testing.assert_close(stored, torch.stack(seen))
    alg.actor_critic.act(stored.flatten(0, 1))
    torch.testing.assert_close(alg.actor_critic.action_mean, alg.storage.mu.flatten(0, 1))      # same weights -> same means: KL = 0


def test_split_k_linear_gradients_match_nn_linear():
    from nightmare_rl_amd.rl.actor_critic import SplitKLinear
    torch.manual_seed(0)
    a, b = SplitKLinear(66, 54), torch.nn.Linear(66, 54)
    b.load_state_dict(a.state_dict())
    x = torch.randn(SplitKLinear.MIN_ROWS, 66, requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    (a(x).tanh().sum()).backward()
    (b(x2).tanh().sum()).backward()
    torch.testing.assert_close(a.weight.grad, b.weight.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(a.bias.grad, b.bias.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(x.grad, x2.grad, rtol=1e-5, atol=1e-6)
    assert list(a.state_dict()) == ["weight", "bias"]
