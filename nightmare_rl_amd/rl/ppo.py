"""PPO: clipped surrogate + clipped value loss, entropy bonus, adaptive-KL learning rate, Adam, grad-norm clipping
(rsl_rl v1.0.2 `algorithms/ppo.py` semantics; reference hyper-parameters envs/nightmare_v3_config.py:111-128).
With torch.distributed initialised, gradients are averaged across ranks with one flat all-reduce per mini-batch and
the KL used for the learning-rate schedule is the global mean, so every rank takes identical steps."""
import os

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.optim as optim

from .fused import FusedCollector, FusedUpdate
from .storage import RolloutStorage


def _dp():
    """Is the update data-parallel? Several ranks - or ONE rank with NM_FORCE_DATA_PARALLEL=1 and an initialised process group: the measurement
    switch that sends a single GPU down the multi-rank code path (all-reduces over RCCL included) so that its kernel trace can be looked at."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("NM_FORCE_DATA_PARALLEL") == "1")


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class PPO:
    def __init__(self, actor_critic, num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95,
                 value_loss_coef=1.0, entropy_coef=0.0, learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True,
                 schedule="fixed", desired_kl=0.01, device="cpu"):
        self.device = device
        self.desired_kl, self.schedule, self.learning_rate = desired_kl, schedule, learning_rate
        self.actor_critic = actor_critic.to(device)
        self.storage = None
        self.optimizer = optim.Adam(self.actor_critic.parameters(), lr=learning_rate)
        self.transition = RolloutStorage.Transition()
        self.clip_param, self.num_learning_epochs, self.num_mini_batches = clip_param, num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef = value_loss_coef, entropy_coef
        self.gamma, self.lam, self.max_grad_norm, self.use_clipped_value_loss = gamma, lam, max_grad_norm, use_clipped_value_loss

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape):
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape, self.device)
        # collection on the hand-written kernels when the networks qualify and the critic sees the actor's observation
        self.fused = self.fused_update = None
        if (critic_obs_shape is None or critic_obs_shape[0] is None) and FusedCollector.supported(self.actor_critic, self.device):
            # the update first: it re-homes the parameters in one flat vector, which the collector then reads
            if FusedUpdate.supported(self.actor_critic, self.device):
                try:
                    self.fused_update = FusedUpdate(self.actor_critic, self.optimizer, self.device, self.learning_rate)
                except Exception as e:     # nm_ppo_create can still refuse (dW tiles per wave, device memory): keep the torch update
                    import warnings
                    warnings.warn(f"fused PPO update unavailable for this network ({e}); using the torch update path")
                    self.fused_update = None
            rank = dist.get_rank() if _world() > 1 else 0     # every rank draws its own action noise
            self.fused = FusedCollector(self.actor_critic, num_envs, self.device, seed=(torch.initial_seed() + 7919 * rank) & 0xFFFFFFFF, update=self.fused_update)
            self.fused.defer_record = True     # one launch per collection step (nm_ppo_record_act); end_rollout() files the last step

    resume_lr_from_checkpoint = False     # runner cfg flag of the same name (not in rsl_rl): see after_load
    update_graph = True                   # fused update on one GPU: replay the mini-batch launches of an update as one HIP graph
    device_permutation = True             # fused update: mini-batch order from nm_ppo_permutation + in-kernel row gather (False: torch.randperm + gathered copies)

    def after_load(self):
        """Parameters / optimizer state were replaced from a checkpoint: let the kernels' copies follow. Like rsl_rl v1.0.2, `learning_rate`
        keeps the configured initial value: the first adaptive-KL mini-batch then writes it (x or / 1.5) over the rate the optimizer's
        param_groups were loaded with. With `resume_lr_from_checkpoint` the schedule continues from the checkpoint's rate instead."""
        lr = self.learning_rate
        if self.optimizer.param_groups:
            if self.resume_lr_from_checkpoint:
                lr = self.learning_rate = float(self.optimizer.param_groups[0]["lr"])
            elif not (self.desired_kl is not None and self.schedule == "adaptive"):
                lr = float(self.optimizer.param_groups[0]["lr"])     # fixed schedule: upstream never rewrites param_groups, the loaded rate is the rate
        if getattr(self, "fused_update", None) is not None:
            self.fused_update.sync(lr)

    def begin_iteration(self, iteration):
        """Once per learning iteration, before the rollout and outside any captured graph: refresh what the collection kernels read."""
        if getattr(self, "fused", None) is not None:
            self.fused.refresh(iteration)

    def test_mode(self):
        self.actor_critic.eval()

    def train_mode(self):
        self.actor_critic.train()

    def act(self, obs, critic_obs):
        if getattr(self, "fused", None) is not None and critic_obs is obs:
            return self.fused.act(obs, self.storage)
        t = self.transition
        t.actions = self.actor_critic.act(obs).detach()
        t.values = self.actor_critic.evaluate(critic_obs).detach()
        t.actions_log_prob = self.actor_critic.get_actions_log_prob(t.actions).detach()
        t.action_mean = self.actor_critic.action_mean.detach()
        t.action_sigma = self.actor_critic.action_std.detach()
        # The env hands out views of persistent device buffers that the next step() overwrites (rsl_rl's envs return fresh
        # tensors): keep what the policy actually saw.
        t.observations = obs.clone()
        t.critic_observations = t.observations if critic_obs is obs else critic_obs.clone()
        return t.actions

    def process_env_step(self, rewards, dones, infos, stats=None, ep=None):
        """stats = (cur_ret, cur_len, fin) device tensors of the runner's episode bookkeeping: with the fused collector they are
        updated in the same launch; returns True then (the caller skips its own bookkeeping). ep = (ep_stats, ep_idx int32, ep_acc):
        the running sum of extras['episode'], also in that launch."""
        if getattr(self, "fused", None) is not None and self.transition.actions is None:
            if stats is None:   # a caller without episode bookkeeping of its own
                if getattr(self, "_own_stats", None) is None:
                    n = rewards.shape[0]
                    self._own_stats = (torch.zeros(n, device=self.device), torch.zeros(n, device=self.device), torch.zeros(3, device=self.device))
                stats = self._own_stats
            self.fused.record(self.storage, rewards, dones, infos.get("time_outs"), self.gamma, *stats, ep=ep)
            return True
        t = self.transition
        t.rewards = rewards.clone()
        t.dones = dones
        if "time_outs" in infos:   # bootstrap the value of envs that merely ran out of time
            t.rewards += self.gamma * torch.squeeze(t.values * infos["time_outs"].unsqueeze(1).to(self.device), 1)
        self.storage.add_transitions(t)
        t.clear()
        self.actor_critic.reset(dones)
        return False

    def reset_collection(self):
        """Empty the rollout storage AND forget a record the fused collector still holds for the next act(): its arguments point into
        the storage rows that are being discarded."""
        self.storage.clear()
        if getattr(self, "fused", None) is not None:
            self.fused.drop_pending()

    def end_rollout(self):
        """After the last process_env_step of a rollout (the runner calls it inside the region it captures into a graph): the fused
        collector files the bookkeeping of the last step, which has no act() after it to ride on."""
        if getattr(self, "fused", None) is not None:
            self.fused.flush()

    def compute_returns(self, last_critic_obs):
        self.end_rollout()
        lv = getattr(self.fused, "last_values", None) if getattr(self, "fused", None) is not None else None
        if lv is not None:
            # the one-launch rollout has evaluated the critic on its last observation already (nm_rollout_args.last_values_dev)
            self.fused.last_values = None
            last_values = lv.view(-1, 1)
        else:
            last_values = self.actor_critic.evaluate(last_critic_obs).detach()
        self.storage.compute_returns(last_values, self.gamma, self.lam)

    def _sync_grads(self):
        if _world() == 1:
            return
        grads = [p.grad for p in self.actor_critic.parameters() if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat)
        flat /= _world()
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n

    def _update_fused(self, defer=None):
        hp = dict(clip=self.clip_param, value_coef=self.value_loss_coef, entropy_coef=self.entropy_coef, clip_value=self.use_clipped_value_loss,
                  desired_kl=self.desired_kl if self.desired_kl is not None else 0.0,
                  adaptive=self.desired_kl is not None and self.schedule == "adaptive", max_grad_norm=self.max_grad_norm)
        fu = self.fused_update
        if self.device_permutation and self.storage.privileged_observations is None:
            # rsl_rl's mini_batch_generator (indices = randperm; obs[batch_idx] ... per mini-batch) without the permuted copies of the rollout:
            # the permutation comes from one small kernel, and the forward / backward kernel gathers its rows through it
            st = self.storage
            B = st.num_envs * st.num_transitions_per_env
            mb = B // self.num_mini_batches
            self._update_count = getattr(self, "_update_count", 0) + 1
            rank = dist.get_rank() if _world() > 1 else 0
            self._perm = fu.permutation(self.num_mini_batches * mb, torch.initial_seed() + 104729 * rank, self._update_count, out=getattr(self, "_perm", None))
            flat = (st.observations.flatten(0, 1), st.actions.flatten(0, 1), st.values.reshape(-1), st.advantages.reshape(-1), st.returns.reshape(-1),
                    st.actions_log_prob.reshape(-1), st.mu.flatten(0, 1), st.sigma.flatten(0, 1))

            def epochs():
                for _ in range(self.num_learning_epochs):
                    for i in range(self.num_mini_batches):
                        rows = self._perm[i * mb:(i + 1) * mb]
                        if _dp():
                            fu.minibatch_data_parallel(*flat, hp=hp, world=_world(), rows=rows)
                        else:
                            fu.minibatch(*flat, hp, rows=rows)

            # All mini-batches of an update read and write fixed addresses (storage rows, the permutation buffer, the flat parameter and
            # moment vectors; learning rate, KL and Adam's step count live on the device): from the third update on, the 2 x epochs x
            # mini-batches launches are ONE HIP graph replay. The permutation above is redrawn every update outside the graph.
            key = (tuple(t.data_ptr() for t in flat), self._perm.data_ptr(), mb, tuple(sorted(hp.items())))
            # With several ranks the graph also holds the all-reduce of every mini-batch (RCCL collectives are capturable; on other
            # backends - the shared-card gloo rehearsal - the update stays on per-launch issue).
            dp_graph = not _dp() or (dist.get_backend() == "nccl" and os.environ.get("NM_DP_UPDATE_GRAPH", "1") != "0")
            if self.update_graph and dp_graph and torch.device(self.device).type == "cuda":
                if getattr(self, "_upd_graph", None) is not None and self._upd_graph[0] == key:
                    self._upd_graph[1].replay()
                    fu.step_count += self.num_learning_epochs * self.num_mini_batches
                elif self._update_count >= 3 and getattr(self, "_upd_graph", None) != "failed":
                    try:
                        torch.cuda.synchronize()
                        g = torch.cuda.CUDAGraph()
                        # under inference_mode like the runner's rollout capture: torch keeps per-generator capture tensors, and whichever
                        # capture of the process comes first decides whether they are inference tensors - in-place updates of those are
                        # allowed in here in both cases, outside only in one
                        # with a process group alive, its watchdog thread may touch the runtime while this thread captures: thread-local
                        # capture mode keeps such a call from invalidating the capture (the collectives themselves are issued by this thread)
                        mode = "thread_local" if _dp() else "global"
                        with torch.inference_mode(), torch.cuda.graph(g, capture_error_mode=mode):
                            epochs()
                        self._upd_graph = (key, g)
                        g.replay()             # the capture executed nothing
                    except Exception as exc:
                        import warnings
                        warnings.warn(f"PPO update graph capture failed ({type(exc).__name__}: {exc}); staying on per-launch updates")
                        self._upd_graph = "failed"
                        epochs()
                else:
                    epochs()
            else:
                epochs()
        else:
            for (obs, _cobs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, _hid, _mask) in \
                    self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs):
                batch = (obs, actions, target_values.reshape(-1), advantages.reshape(-1), returns.reshape(-1), old_logp.reshape(-1), old_mu, old_sigma)
                if _dp():
                    fu.minibatch_data_parallel(*batch, hp=hp, world=_world())
                else:
                    fu.minibatch(*batch, hp)
        if defer is not None:
            # the caller reads the update's statistics later (runner: one iteration later, while the next rollout runs): no host
            # synchronisation here. The learning rate, KL and Adam's step count that matter live on the device; apply_state() brings the
            # host's copies (logging, checkpoints) up to date
            fu.snapshot_state(defer)
            self.reset_collection()
            return None
        st = fu.read_state()                      # the one host synchronisation of the update
        self.reset_collection()
        return self.apply_state(st)

    def apply_state(self, st):
        """Host-side mirror of what the fused update left on the device (st: FusedUpdate.read_state() / state_from()); returns the
        mean value and surrogate losses of the update like update() does."""
        self.learning_rate, self.last_kl = st["lr"], st["kl"]
        for g in self.optimizer.param_groups:
            g["lr"] = self.learning_rate
        n = max(st["minibatches"], 1.0)
        return st["value_loss_sum"] / n, st["surrogate_loss_sum"] / n

    def update(self, defer=None):
        """rsl_rl PPO.update. defer (fused path only): a 9-float device tensor that receives the update's statistics instead of a host read."""
        if getattr(self, "fused_update", None) is not None:
            return self._update_fused(defer)
        v_sum = torch.zeros((), device=self.device)
        s_sum = torch.zeros((), device=self.device)
        gen = self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs)
        for (obs, cobs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, _hid, _mask) in gen:
            self.actor_critic.act(obs)
            logp = self.actor_critic.get_actions_log_prob(actions)
            value = self.actor_critic.evaluate(cobs)
            mu, sigma, entropy = self.actor_critic.action_mean, self.actor_critic.action_std, self.actor_critic.entropy
            if self.desired_kl is not None and self.schedule == "adaptive":
                with torch.inference_mode():
                    kl = torch.sum(torch.log(sigma / old_sigma + 1.0e-5) + (old_sigma.square() + (old_mu - mu).square()) / (2.0 * sigma.square()) - 0.5, dim=-1)
                    kl_mean = kl.mean()
                    if _world() > 1:
                        dist.all_reduce(kl_mean)
                        kl_mean /= _world()
                    kl_mean = float(kl_mean)   # the one host sync per mini-batch the schedule needs
                    self.last_kl = kl_mean
                if kl_mean > self.desired_kl * 2.0:
                    self.learning_rate = max(1e-5, self.learning_rate / 1.5)
                elif 0.0 < kl_mean < self.desired_kl / 2.0:
                    self.learning_rate = min(1e-2, self.learning_rate * 1.5)
                for g in self.optimizer.param_groups:
                    g["lr"] = self.learning_rate
            adv = advantages.squeeze(-1)
            ratio = torch.exp(logp - old_logp.squeeze(-1))
            surrogate_loss = torch.max(-adv * ratio, -adv * ratio.clamp(1.0 - self.clip_param, 1.0 + self.clip_param)).mean()
            if self.use_clipped_value_loss:
                v_clip = target_values + (value - target_values).clamp(-self.clip_param, self.clip_param)
                value_loss = torch.max((value - returns).square(), (v_clip - returns).square()).mean()
            else:
                value_loss = (returns - value).square().mean()
            loss = surrogate_loss + self.value_loss_coef * value_loss - self.entropy_coef * entropy.mean()
            self.optimizer.zero_grad()
            loss.backward()
            self._sync_grads()
            nn.utils.clip_grad_norm_(self.actor_critic.parameters(), self.max_grad_norm)
            self.optimizer.step()
            v_sum += value_loss.detach()
            s_sum += surrogate_loss.detach()
        n = self.num_learning_epochs * self.num_mini_batches
        self.storage.clear()
        return float(v_sum / n), float(s_sum / n)
