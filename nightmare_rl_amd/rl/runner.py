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
from .ppo import PPO, _dp

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
        # not in rsl_rl: continue the adaptive-KL schedule from a checkpoint's learning rate instead of the configured one
        self.alg.resume_lr_from_checkpoint = bool(self.cfg.get("resume_lr_from_checkpoint", False))
        self.num_steps_per_env, self.save_interval = self.cfg["num_steps_per_env"], self.cfg["save_interval"]
        # the reference env announces num_privileged_obs = num_obs (env.py:34) but hands out no privileged observations (:317-319):
        # the critic then sees the actor's observation, and a second copy of it in the storage would only cost bandwidth
        has_priv = env.get_privileged_observations() is not None
        self.alg.init_storage(env.num_envs, self.num_steps_per_env, [env.num_obs], [env.num_privileged_obs if has_priv else None], [env.num_actions])
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
        T = self.num_steps_per_env
        fin = torch.zeros(3, device=dev)         # finished episodes this iteration: sum of returns, sum of lengths, count
        ep_acc = None                            # running sum of extras['episode'] over the steps of one rollout
        state = {"obs": obs, "critic_obs": critic_obs, "ep_keys": None}

        def rollout():
            """One rollout of T steps. Nothing in here reads device memory from the host, shapes and buffers are the same every
            iteration, so on a GPU the whole thing (policy, sampling, env kernels, storage writes, statistics) is captured once
            into a HIP graph and replayed."""
            nonlocal ep_acc
            o, co = state["obs"], state["critic_obs"]
            fin.zero_()
            if ep_acc is not None:
                ep_acc.zero_()
            if one_launch:
                # the whole loop below inside the env's own waves: policy, step, record, K times, no launch in between (nm_rollout)
                if state.get("ep_idx") is None:
                    keys = sorted(env.extras["episode"])
                    state["ep_keys"] = keys
                    state["ep_idx"] = torch.tensor([env._stat_names.index(k[4:]) for k in keys], device=dev, dtype=torch.int32)
                    state["ep_idx64"] = state["ep_idx"].long()
                if ep_acc is None:
                    ep_acc = torch.zeros(len(state["ep_keys"]), device=dev)
                o = alg.fused.rollout(env, alg.storage, T, alg.gamma, cur_ret, cur_len, fin, ep=(state["ep_idx"], ep_acc))
                state["obs"] = state["critic_obs"] = o
                return
            for _ in range(T):
                actions = alg.act(o, co)
                o, priv_, rewards, dones, infos = env.step(actions)
                o, rewards, dones = o.to(dev), rewards.to(dev), dones.to(dev)
                co = priv_.to(dev) if priv_ is not None else o
                ep = None
                if "episode" in infos and hasattr(env, "_ep_stats") and hasattr(env, "_stat_names"):
                    keys = sorted(infos["episode"])
                    if state.get("ep_idx") is None:
                        state["ep_idx"] = torch.tensor([env._stat_names.index(k[4:]) for k in keys], device=dev, dtype=torch.int32)
                        state["ep_idx64"] = state["ep_idx"].long()
                    if ep_acc is None:
                        ep_acc = torch.zeros(len(keys), device=dev)
                    state["ep_keys"] = keys
                    ep = (env._ep_stats, state["ep_idx"], ep_acc)
                fused = alg.process_env_step(rewards, dones, infos, stats=(cur_ret, cur_len, fin), ep=ep)
                if not fused:
                    cur_ret.add_(rewards)
                    cur_len.add_(1)
                    d = (dones > 0).float()
                    fin.add_(torch.stack([(cur_ret * d).sum(), (cur_len * d).sum(), d.sum()]))
                    cur_ret.mul_(1 - d)
                    cur_len.mul_(1 - d)
                if "episode" in infos and not (fused and ep is not None):    # the fused record launch has added it already
                    if ep is not None:
                        e = env._ep_stats.index_select(0, state["ep_idx64"])
                    else:
                        keys = sorted(infos["episode"])
                        e = torch.stack([infos["episode"][k].float() for k in keys])
                        state["ep_keys"] = keys
                    if ep_acc is None:
                        ep_acc = torch.zeros_like(e)
                    ep_acc.add_(e)
            if hasattr(alg, "end_rollout"):
                alg.end_rollout()
            # The rollout starts from a buffer of its own: a captured graph reads and writes fixed addresses, and an env that
            # alternates its observation buffers ends an odd-length rollout in the buffer the next replay would NOT read first.
            if on_gpu and priv is None:
                if state.get("stage") is None:
                    state["stage"] = torch.empty_like(o)
                state["stage"].copy_(o)
                o = co = state["stage"]
            state["obs"], state["critic_obs"] = o, co

        on_gpu = torch.device(dev).type == "cuda"
        one_launch = bool(self.cfg.get("fused_rollout", True)) and getattr(alg, "fused", None) is not None and priv is None \
            and alg.fused.can_rollout(env) and T <= int(env.max_episode_length)
        self.rollout_mode = "one launch (nm_rollout)" if one_launch else "per-step launches"
        want_graph = not one_launch and bool(self.cfg.get("graph_rollout", True)) and torch.device(dev).type == "cuda" and hasattr(env, "_h") \
            and not getattr(env, "add_noise", False) and not getattr(getattr(env.cfg, "viewer", None), "record_states", False)
        graph = None
        ev0, ev1 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if on_gpu else (None, None)
        # Pipelined logging (fused update): rsl_rl's runner reads the iteration's statistics from the device after every update
        # (OnPolicyRunner.log; caller reference train.py:54) - a handful of host synchronisations during which the GPU idles (measured:
        # 0.2 ms of a 7.4 ms iteration, 0.5 ms under the profiler). Here the statistics of iteration i are copied, stream-ordered, into a pinned host buffer and READ
        # after iteration i + 1 has been enqueued; iterations that save a checkpoint, and the last one, are drained at once, so a
        # checkpoint still holds exactly the state after its iteration. cfg pipeline_logging=False restores the synchronous loop.
        # With several ranks the iteration's statistics that span ranks (finished episodes, mean step reward) are all-reduced ON THE DEVICE
        # inside the snapshot - one small collective per iteration, enqueued like everything else - and read one iteration later as well.
        pipe = on_gpu and getattr(alg, "fused_update", None) is not None and bool(self.cfg.get("pipeline_logging", True))
        self.logging_mode = "pipelined (read one iteration later)" if pipe else "synchronous"
        kSnap = 9 + 3 + 1 + 1                     # PPO state | finished episodes | mean step reward | mean action std | extras['episode'] sums follow
        slots = []                                # two sets of (pinned host buffer, events) used alternately
        snap_dev = torch.zeros(64, device=dev) if pipe else None
        pending = []                              # iterations whose statistics have not been read yet

        def emit(it, collection_time, learn_time, mean_value_loss, mean_surrogate_loss, kl, f, ep_vals, keys, ep_n, action_std, lr):
            nonlocal mean_ret, mean_len
            if f[2] > 0:
                mean_ret, mean_len = f[0] / f[2], f[1] / f[2]
            steps = self.num_steps_per_env * N * _world()
            self.tot_timesteps += steps
            self.tot_time += collection_time + learn_time
            rec = dict(it=it, fps=steps / (collection_time + learn_time), collection_time=collection_time, learn_time=learn_time,
                       value_loss=mean_value_loss, surrogate_loss=mean_surrogate_loss, kl=kl, mean_reward=mean_ret, mean_episode_length=mean_len, mean_step_reward=f[3],
                       action_std=action_std, lr=lr, total_timesteps=self.tot_timesteps)
            if ep_vals is not None and keys:
                for k, v in zip(keys, ep_vals):
                    rec["episode/" + k] = v / max(ep_n, 1)
            self.history.append(rec)
            if log:
                with open(os.path.join(self.log_dir, "progress.jsonl"), "a") as fh:
                    fh.write(json.dumps(rec) + "\n")
                print(f"it {it:5d}/{tot_iter} | {rec['fps']:10.0f} steps/s (collect {collection_time:.3f}s learn {learn_time:.3f}s) | "
                      f"value {mean_value_loss:.4f} surrogate {mean_surrogate_loss:.4f} | reward {mean_ret:8.3f} len {mean_len:7.1f} | "
                      f"std {rec['action_std']:.3f} lr {rec['lr']:.2e}", flush=True)
                if it % self.save_interval == 0:
                    self.save(os.path.join(self.log_dir, f"model_{it}.pt"))

        def drain(upto_all):
            """Read and log the pending iterations (all of them, or all but the newest)."""
            while pending and (upto_all or len(pending) > 1):
                it_, sl, keys, ep_n, nxt = pending.pop(0)
                host, e0, e1, e2 = sl
                e2.synchronize()
                v = host.tolist()
                st = alg.fused_update.state_from(v[:9])
                vl, sl_ = alg.apply_state(st)
                # the iteration's time on the GPU: from its first launch to the first launch of the next iteration (to its own last one
                # if nothing follows yet); the device is busy without a gap in between
                end = nxt[1] if nxt is not None else e2
                if nxt is not None:
                    end.synchronize()
                total = e0.elapsed_time(end) * 1e-3
                collection = e0.elapsed_time(e1) * 1e-3
                emit(it_, collection, max(total - collection, 0.0), vl, sl_, st["kl"], v[9:13], v[kSnap:kSnap + len(keys)] if keys else None, keys, ep_n, v[13], st["lr"])

        # with pipelined logging the host mirrors (Adam step count, learning rate) lag the device by one iteration until the
        # queue is drained - so it IS drained on every way out of the loop (exception, KeyboardInterrupt), and save() drains first
        self._drain_all = (lambda: drain(True)) if pipe else None
        try:
            for it in range(self.current_learning_iteration, tot_iter):
                start = time.time()
                if hasattr(alg, "begin_iteration"):
                    alg.begin_iteration(it)
                if pipe:
                    if len(slots) < 2:
                        slots.append((torch.zeros(64, dtype=torch.float32).pin_memory(), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
                                      torch.cuda.Event(enable_timing=True)))
                    sl = slots[(it - self.current_learning_iteration) % 2]
                    ev0, ev1 = sl[1], sl[2]
                    if pending:
                        pending[-1][4] = sl            # the previous iteration ends where this one starts
                if on_gpu:
                    ev0.record()
                with torch.inference_mode():
                    if graph is not None:
                        graph.replay()
                        alg.storage.step = T         # what the captured add_transitions calls did on the host side
                        env.common_step_counter += T
                    elif want_graph and it > self.current_learning_iteration and ep_acc is not None:
                        # the first iteration ran eagerly (lazy initialisations done, buffers exist): capture the next one
                        try:
                            torch.cuda.synchronize()
                            g = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(g):
                                rollout()
                            graph = g               # the capture did not execute anything: run this iteration from the graph
                            alg.storage.clear()
                            graph.replay()
                            alg.storage.step = T
                        except Exception as exc:     # keep training on the eager path
                            want_graph = False
                            alg.reset_collection() if hasattr(alg, "reset_collection") else alg.storage.clear()
                            if log:
                                print(f"rollout graph capture failed ({type(exc).__name__}: {exc}); staying on the eager path", flush=True)
                            rollout()
                    else:
                        rollout()
                    obs, critic_obs = state["obs"], state["critic_obs"]
                    ep_stats, ep_n = (ep_acc, T) if ep_acc is not None else (None, 0)
                    infos = env.extras
                    if on_gpu:
                        ev1.record()             # the rollout is asynchronous: its device time is read after the update has synchronised
                    collection_time = time.time() - start
                    step_rew = alg.storage.rewards.mean()
                    alg.compute_returns(critic_obs)
                if pipe:
                    alg.update(defer=snap_dev[:9])
                    with torch.inference_mode():
                        keys = list(state["ep_keys"]) if (ep_stats is not None and state["ep_keys"]) else []
                        snap_dev[9:12].copy_(fin)
                        snap_dev[12].copy_(step_rew)
                        if _world() > 1 or _dp():
                            dist.all_reduce(snap_dev[9:13])       # sums over the ranks, still on the device; the mean step reward is divided below
                            snap_dev[12].div_(_world())
                        snap_dev[13].copy_(alg.actor_critic.std.detach().mean())
                        if keys:
                            snap_dev[kSnap:kSnap + len(keys)].copy_(ep_stats)
                        sl[0].copy_(snap_dev, non_blocking=True)
                    sl[3].record()
                    pending.append([it, sl, keys, ep_n, None])
                    # a checkpoint must hold the state after ITS iteration: drain before the next one is enqueued; else lag by one iteration
                    drain(upto_all=(log and it % self.save_interval == 0) or it == tot_iter - 1)
                    continue
                mean_value_loss, mean_surrogate_loss = alg.update()
                kl = getattr(alg, "last_kl", float("nan"))
                total_time = time.time() - start
                if on_gpu:
                    torch.cuda.synchronize()
                    total_time = time.time() - start
                    collection_time = ev0.elapsed_time(ev1) * 1e-3
                learn_time = max(total_time - collection_time, 0.0)
                if _world() > 1:
                    dist.all_reduce(fin)
                if _world() > 1:
                    step_rew = step_rew.clone()     # made under inference_mode above
                    dist.all_reduce(step_rew)
                    step_rew = step_rew / _world()
                f = fin.tolist() + [float(step_rew)]   # the one host read of episode statistics per iteration
                emit(it, collection_time, learn_time, mean_value_loss, mean_surrogate_loss, kl, f,
                     ep_stats.tolist() if (ep_stats is not None and state["ep_keys"]) else None, state["ep_keys"], ep_n,
                     float(alg.actor_critic.std.detach().mean()), alg.learning_rate)
        finally:
            try:
                if pipe and pending:
                    drain(True)
            finally:
                self._drain_all = None
        self.current_learning_iteration += int(num_learning_iterations)
        if log:
            self.save(os.path.join(self.log_dir, f"model_{self.current_learning_iteration}.pt"))

    def save(self, path, infos=None):
        fu = getattr(self.alg, "fused_update", None)
        drain_all, self._drain_all = getattr(self, "_drain_all", None), None      # (re-entered from the drain's own emit(): nothing left to do)
        if drain_all is not None:
            try:
                drain_all()                       # a checkpoint holds the state after every iteration that has been enqueued
            finally:
                self._drain_all = drain_all
        if fu is not None:
            fu.publish_step()                     # Adam's step count into the optimizer state that is saved below
        torch.save({"model_state_dict": self.alg.actor_critic.state_dict(), "optimizer_state_dict": self.alg.optimizer.state_dict(),
                    "iter": self.current_learning_iteration, "infos": infos}, path)

    def load(self, path, load_optimizer=True):
        d = torch.load(path, map_location=self.device)
        self.alg.actor_critic.load_state_dict(d["model_state_dict"])
        if load_optimizer:
            self.alg.optimizer.load_state_dict(d["optimizer_state_dict"])
        self.current_learning_iteration = d["iter"]
        if hasattr(self.alg, "after_load"):
            self.alg.after_load()
        return d["infos"]

    def get_inference_policy(self, device=None):
        self.alg.actor_critic.eval()
        if device is not None:
            self.alg.actor_critic.to(device)
        return self.alg.actor_critic.act_inference
