"""OnPolicyRunner with the surface of rsl_rl v1.0.2's runner (reference train.py:40,52,54): construct from
(env, train_cfg dict, log_dir, device), `.learn(num_learning_iterations, init_at_random_ep_len)`, `.save/.load`
(`model_<it>.pt` with 'model_state_dict' / 'optimizer_state_dict' / 'iter' / 'infos'), `.get_inference_policy()`.

GPU-first differences: the rollout loop never synchronises with the host (episode statistics accumulate in device
tensors and are read once per iteration); rollout storage lives in HBM; with torch.distributed initialised every rank
owns a shard of the envs, steps are identical on all ranks (gradient + KL + advantage statistics all-reduced) and only
rank 0 logs and saves."""
import json
import os
import time

import torch
import torch.distributed as dist

from .actor_critic import ActorCritic
from .ppo import PPO

_CLASSES = {"ActorCritic": ActorCritic, "PPO": PPO}


def _rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class OnPolicyRunner:
    def __init__(self, env, train_cfg, log_dir=None, device="cpu"):
        self.cfg, self.alg_cfg, self.policy_cfg = train_cfg["runner"], train_cfg["algorithm"], train_cfg["policy"]
        self.device, self.env = device, env
        num_critic_obs = env.num_privileged_obs if env.num_privileged_obs is not None else env.num_obs
        ac = _CLASSES[self.cfg["policy_class_name"]](env.num_obs, num_critic_obs, env.num_actions, **self.policy_cfg).to(device)
        if _world() > 1:   # identical initial weights on every rank
            for p in ac.parameters():
                dist.broadcast(p.data, 0)
        self.alg = _CLASSES[self.cfg["algorithm_class_name"]](ac, device=device, **self.alg_cfg)
        self.num_steps_per_env, self.save_interval = self.cfg["num_steps_per_env"], self.cfg["save_interval"]
        self.alg.init_storage(env.num_envs, self.num_steps_per_env, [env.num_obs], [env.num_privileged_obs], [env.num_actions])
        self.log_dir = log_dir
        self.tot_timesteps, self.tot_time, self.current_learning_iteration = 0, 0.0, 0
        self.history = []
        self.env.reset()

    def learn(self, num_learning_iterations, init_at_random_ep_len=False):
        env, alg, dev = self.env, self.alg, self.device
        if init_at_random_ep_len:
            env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
        obs = env.get_observations().to(dev)
        priv = env.get_privileged_observations()
        critic_obs = priv.to(dev) if priv is not None else obs
        alg.actor_critic.train()
        N = env.num_envs
        cur_ret = torch.zeros(N, device=dev)
        cur_len = torch.zeros(N, device=dev)
        mean_ret = mean_len = float("nan")
        tot_iter = self.current_learning_iteration + int(num_learning_iterations)
        log = self.log_dir is not None and _rank() == 0
        if log:
            os.makedirs(self.log_dir, exist_ok=True)
        for it in range(self.current_learning_iteration, tot_iter):
            start = time.time()
            fin = torch.zeros(3, device=dev)     # finished episodes this iteration: sum of returns, sum of lengths, count
            ep_stats, ep_n = None, 0
            with torch.inference_mode():
                for _ in range(self.num_steps_per_env):
                    actions = alg.act(obs, critic_obs)
                    obs, priv, rewards, dones, infos = env.step(actions)
                    obs, rewards, dones = obs.to(dev), rewards.to(dev), dones.to(dev)
                    critic_obs = priv.to(dev) if priv is not None else obs
                    alg.process_env_step(rewards, dones, infos)
                    cur_ret += rewards
                    cur_len += 1
                    d = (dones > 0).float()
                    fin += torch.stack([(cur_ret * d).sum(), (cur_len * d).sum(), d.sum()])
                    cur_ret *= 1 - d
                    cur_len *= 1 - d
                    if "episode" in infos:
                        e = torch.stack([infos["episode"][k].float() for k in sorted(infos["episode"])])
                        ep_stats = e.clone() if ep_stats is None else ep_stats + e
                        ep_n += 1
                collection_time = time.time() - start
                start = time.time()
                step_rew = alg.storage.rewards.mean()
                alg.compute_returns(critic_obs)
            mean_value_loss, mean_surrogate_loss = alg.update()
            kl = getattr(alg, "last_kl", float("nan"))
            learn_time = time.time() - start
            if _world() > 1:
                dist.all_reduce(fin)
            if _world() > 1:
                step_rew = step_rew.clone()     # made under inference_mode above
                dist.all_reduce(step_rew)
                step_rew = step_rew / _world()
            f = fin.tolist() + [float(step_rew)]   # the one host read of episode statistics per iteration
            if f[2] > 0:
                mean_ret, mean_len = f[0] / f[2], f[1] / f[2]
            steps = self.num_steps_per_env * N * _world()
            self.tot_timesteps += steps
            self.tot_time += collection_time + learn_time
            rec = dict(it=it, fps=steps / (collection_time + learn_time), collection_time=collection_time, learn_time=learn_time,
                       value_loss=mean_value_loss, surrogate_loss=mean_surrogate_loss, kl=kl, mean_reward=mean_ret, mean_episode_length=mean_len, mean_step_reward=f[3],
                       action_std=float(alg.actor_critic.std.detach().mean()), lr=alg.learning_rate, total_timesteps=self.tot_timesteps)
            if ep_stats is not None and "episode" in infos:
                for k, v in zip(sorted(infos["episode"]), (ep_stats / max(ep_n, 1)).tolist()):
                    rec["episode/" + k] = v
            self.history.append(rec)
            if log:
                with open(os.path.join(self.log_dir, "progress.jsonl"), "a") as fh:
                    fh.write(json.dumps(rec) + "\n")
                print(f"it {it:5d}/{tot_iter} | {rec['fps']:10.0f} steps/s (collect {collection_time:.3f}s learn {learn_time:.3f}s) | "
                      f"value {mean_value_loss:.4f} surrogate {mean_surrogate_loss:.4f} | reward {mean_ret:8.3f} len {mean_len:7.1f} | "
                      f"std {rec['action_std']:.3f} lr {rec['lr']:.2e}", flush=True)
                if it % self.save_interval == 0:
                    self.save(os.path.join(self.log_dir, f"model_{it}.pt"))
        self.current_learning_iteration += int(num_learning_iterations)
        if log:
            self.save(os.path.join(self.log_dir, f"model_{self.current_learning_iteration}.pt"))

    def save(self, path, infos=None):
        torch.save({"model_state_dict": self.alg.actor_critic.state_dict(), "optimizer_state_dict": self.alg.optimizer.state_dict(),
                    "iter": self.current_learning_iteration, "infos": infos}, path)

    def load(self, path, load_optimizer=True):
        d = torch.load(path, map_location=self.device)
        self.alg.actor_critic.load_state_dict(d["model_state_dict"])
        if load_optimizer:
            self.alg.optimizer.load_state_dict(d["optimizer_state_dict"])
        self.current_learning_iteration = d["iter"]
        return d["infos"]

    def get_inference_policy(self, device=None):
        self.alg.actor_critic.eval()
        if device is not None:
            self.alg.actor_critic.to(device)
        return self.alg.actor_critic.act_inference
