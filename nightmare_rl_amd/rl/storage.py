"""RolloutStorage: [steps, envs, ...] buffers resident on the device; GAE(lambda) returns; shuffled mini-batches
(rsl_rl v1.0.2 `storage/rollout_storage.py` semantics). The backward GAE scan runs in one HIP kernel (nm_gae) on a
GPU and in torch on the host (tests)."""
import ctypes as C

import torch
import torch.distributed as dist

from ..distributed import global_advantage_stats


class RolloutStorage:
    class Transition:
        def __init__(self):
            self.clear()

        def clear(self):
            self.observations = self.critic_observations = self.actions = self.rewards = self.dones = None
            self.values = self.actions_log_prob = self.action_mean = self.action_sigma = None

    def __init__(self, num_envs, num_transitions_per_env, obs_shape, privileged_obs_shape, actions_shape, device="cpu"):
        self.device = device
        self.num_envs, self.num_transitions_per_env = num_envs, num_transitions_per_env
        T, N = num_transitions_per_env, num_envs
        z = lambda *s: torch.zeros(T, N, *s, device=device)
        self.observations = z(*obs_shape)
        self.privileged_observations = z(*privileged_obs_shape) if privileged_obs_shape and privileged_obs_shape[0] is not None else None
        self.rewards, self.values, self.returns, self.advantages, self.actions_log_prob = z(1), z(1), z(1), z(1), z(1)
        self.actions, self.mu, self.sigma = z(*actions_shape), z(*actions_shape), z(*actions_shape)
        self.dones = torch.zeros(T, N, 1, device=device, dtype=torch.uint8)
        self.step = 0

    def add_transitions(self, t):
        if self.step >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        s = self.step
        self.observations[s].copy_(t.observations)
        if self.privileged_observations is not None:
            self.privileged_observations[s].copy_(t.critic_observations)
        self.actions[s].copy_(t.actions)
        self.rewards[s].copy_(t.rewards.view(-1, 1))
        self.dones[s].copy_(t.dones.view(-1, 1))
        self.values[s].copy_(t.values)
        self.actions_log_prob[s].copy_(t.actions_log_prob.view(-1, 1))
        self.mu[s].copy_(t.action_mean)
        self.sigma[s].copy_(t.action_sigma)
        self.step += 1

    def clear(self):
        self.step = 0

    def compute_returns(self, last_values, gamma, lam):
        if self.rewards.is_cuda:
            from .. import _lib
            L = _lib.load()
            T, N = self.num_transitions_per_env, self.num_envs
            stream = C.c_void_p(torch.cuda.current_stream(self.rewards.device).cuda_stream)
            lv = last_values.contiguous().view(-1).float()
            one = not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
            with torch.cuda.device(self.rewards.device):
                if one:
                    # returns, advantages and their normalisation in two launches (nm_gae_advantages); with several ranks the
                    # statistics are all-reduced below
                    if getattr(self, "_gae_scratch", None) is None:
                        self._gae_scratch = torch.zeros(2 * ((N + 255) // 256), device=self.rewards.device)
                    _lib.check(L.nm_gae_advantages(self.rewards.data_ptr(), self.values.data_ptr(), self.dones.data_ptr(), lv.data_ptr(), T, N, float(gamma),
                                                   float(lam), self.returns.data_ptr(), self.advantages.data_ptr(), self._gae_scratch.data_ptr(), 1, stream))
                    return
                _lib.check(L.nm_gae(self.rewards.data_ptr(), self.values.data_ptr(), self.dones.data_ptr(), lv.data_ptr(), T, N,
                                    float(gamma), float(lam), self.returns.data_ptr(), stream))
        else:
            adv = torch.zeros_like(last_values)
            for s in reversed(range(self.num_transitions_per_env)):
                nxt = last_values if s == self.num_transitions_per_env - 1 else self.values[s + 1]
                live = 1.0 - self.dones[s].float()
                delta = self.rewards[s] + live * gamma * nxt - self.values[s]
                adv = delta + live * gamma * lam * adv
                self.returns[s] = adv + self.values[s]
        # in place: the update's captured graph (rl/ppo.py) reads the advantages at a fixed address
        torch.sub(self.returns, self.values, out=self.advantages)
        mean, std = global_advantage_stats(self.advantages)   # over ALL ranks' envs x steps
        self.advantages.sub_(mean).div_(std + 1e-8)

    def get_statistics(self):
        done = self.dones.clone()
        done[-1] = 1
        flat = done.permute(1, 0, 2).reshape(-1, 1)
        idx = torch.cat((flat.new_tensor([-1], dtype=torch.int64), flat.nonzero(as_tuple=False)[:, 0]))
        lengths = idx[1:] - idx[:-1]
        return lengths.float().mean(), self.rewards.mean()

    def mini_batch_generator(self, num_mini_batches, num_epochs=8):
        B = self.num_envs * self.num_transitions_per_env
        mb = B // num_mini_batches
        idx = torch.randperm(num_mini_batches * mb, device=self.device)
        obs = self.observations.flatten(0, 1)
        cobs = self.privileged_observations.flatten(0, 1) if self.privileged_observations is not None else obs
        flat = [t.flatten(0, 1) for t in (self.actions, self.values, self.advantages, self.returns, self.actions_log_prob, self.mu, self.sigma)]
        # one gather of the whole buffer per update instead of one per mini-batch and epoch: the permutation is drawn once
        # (as upstream), so every epoch sees the same mini-batches - contiguous slices of the permuted copy
        obs_p = obs[idx]
        cobs_p = cobs[idx] if self.privileged_observations is not None else obs_p
        flat_p = [t[idx] for t in flat]
        for _ in range(num_epochs):
            for i in range(num_mini_batches):
                sl = slice(i * mb, (i + 1) * mb)
                yield (obs_p[sl], cobs_p[sl]) + tuple(t[sl] for t in flat_p) + ((None, None), None)
