"""Rollout collection on hand-written kernels: what rsl_rl v1.0.2's `PPO.act` + `PPO.process_env_step` do per step (actor mean,
critic value, Normal sample, log-probability, transition record, time-out bootstrap, episode bookkeeping; caller reference
train.py:54), writing straight into the rollout storage:
  * networks of the reference's shape, next to a FusedUpdate: ONE launch per step - `nm_ppo_record_act`: the bookkeeping of the previous
    step (`nm_ppo_record`'s arithmetic) at the head of `nm_ppo_act` (register-resident forward of the merged actor+critic network from
    the update's own packed weights + sampling head); the first act of a rollout is a plain `nm_ppo_act`, the last step's bookkeeping
    one `nm_ppo_record` (`flush`, from PPO.end_rollout);
  * other qualifying networks: THREE - the fused actor+critic forward on the matrix cores (`nm_policy_forward` on one merged network:
    the two MLPs side by side, block-diagonal hidden layers), `nm_ppo_sample`, `nm_ppo_record`; the packed copy of the parameters is
    refreshed once per iteration by `refresh()`, outside any captured graph.

Used by PPO when the networks qualify (`FusedCollector.supported`); otherwise PPO keeps its torch path (host tests, other
activations)."""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from ..policy import PackedMLP


class FusedCollector:
    @staticmethod
    def supported(ac, device):
        if torch.device(device).type != "cuda" or getattr(ac, "activation_name", None) != "elu":
            return False
        a = [m for m in ac.actor if isinstance(m, nn.Linear)]
        c = [m for m in ac.critic if isinstance(m, nn.Linear)]
        if len(a) != len(c) or len(a) > 4 or a[0].in_features != c[0].in_features or c[-1].out_features != 1 or a[-1].out_features > 32:
            return False
        return all(x.out_features + y.out_features <= 256 for x, y in zip(a, c)) and a[0].in_features <= 256

    def __init__(self, ac, num_envs, device, seed=0, update=None):
        self.last_values = None          # set by rollout(): the critic's value of the rollout's last observation, for PPO.compute_returns
        """update: the FusedUpdate of the same networks, if there is one. When its handle has the compiled fast path, `act` is ONE
        launch (nm_ppo_act: forward from the update's own packed weights + sampling) and `refresh` has nothing to repack."""
        self.ac, self.device, self.N = ac, torch.device(device), int(num_envs)
        self.update = update if (update is not None and update.has_fast_path) else None
        self.a_lin = [m for m in ac.actor if isinstance(m, nn.Linear)]
        self.c_lin = [m for m in ac.critic if isinstance(m, nn.Linear)]
        self.A = self.a_lin[-1].out_features
        dims = [self.a_lin[0].in_features] + [x.out_features + y.out_features for x, y in zip(self.a_lin, self.c_lin)]
        self.net = PackedMLP(dims, self.device)
        self.out = torch.empty(self.N, self.A + 1, device=self.device)
        self.iter_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.seed = int(seed)
        self._L = _lib.load()
        self._pending = None          # arguments of a record() that waits for the next act() (defer_record)
        self.defer_record = False     # set by PPO: record(s) rides at the head of act(s + 1); PPO.end_rollout / compute_returns flush the last one
        self.refresh(0)

    @torch.no_grad()
    def refresh(self, iteration):
        """Merge the current actor / critic parameters into the packed network and set the iteration the action noise is keyed by.
        Call once per learning iteration, outside graph capture."""
        if self.update is None:
            ws, bs = [], []
            for l, (x, y) in enumerate(zip(self.a_lin, self.c_lin)):
                ws.append(torch.cat([x.weight, y.weight], 0) if l == 0 else torch.block_diag(x.weight, y.weight))
                bs.append(torch.cat([x.bias, y.bias], 0))
            self.net.load(ws, bs)
        self.std = self.ac.std.detach().contiguous()
        self.iter_dev.fill_(int(iteration))

    def act(self, obs, storage):
        """One collection step: actions for `obs`, with the whole transition filed into step `storage.step` of the storage."""
        s = storage.step
        if s >= storage.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        obs = obs if (obs.dtype == torch.float32 and obs.is_contiguous()) else obs.contiguous().float()
        if self.update is not None:
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            if self._pending is not None:      # the previous step's bookkeeping rides at the head of this launch
                rec, self._pending = self._pending, None
                _lib.check(self._L.nm_ppo_record_act(self.update._h, *rec, self.update.flat.data_ptr(), obs.data_ptr(), self.N, self.seed,
                                                     self.iter_dev.data_ptr(), s, storage.actions[s].data_ptr(), storage.actions_log_prob[s].data_ptr(),
                                                     storage.values[s].data_ptr(), storage.mu[s].data_ptr(), storage.sigma[s].data_ptr(),
                                                     storage.observations[s].data_ptr(), stream))
                return storage.actions[s]
            _lib.check(self._L.nm_ppo_act(self.update._h, self.update.flat.data_ptr(), obs.data_ptr(), self.N, self.seed, self.iter_dev.data_ptr(), s,
                                          storage.actions[s].data_ptr(), storage.actions_log_prob[s].data_ptr(), storage.values[s].data_ptr(),
                                          storage.mu[s].data_ptr(), storage.sigma[s].data_ptr(), storage.observations[s].data_ptr(), stream))
            return storage.actions[s]
        self.net.forward(obs, out=self.out)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._L.nm_ppo_sample(self.out.data_ptr(), self.std.data_ptr(), obs.data_ptr(), self.N, self.A, obs.shape[1], self.seed,
                                         self.iter_dev.data_ptr(), s, storage.actions[s].data_ptr(), storage.actions_log_prob[s].data_ptr(),
                                         storage.values[s].data_ptr(), storage.mu[s].data_ptr(), storage.sigma[s].data_ptr(),
                                         storage.observations[s].data_ptr(), stream))
        return storage.actions[s]

    def record(self, storage, rewards, dones, time_outs, gamma, cur_ret, cur_len, fin, ep=None):
        """ep = (ep_stats [K] f32, ep_idx [n] int32, ep_acc [n] f32) device tensors: ep_acc += ep_stats[ep_idx] in the same launch."""
        s = storage.step
        # the kernel reads raw pointers: float32 rewards / time_outs, int64 dones, contiguous, on this device. An env in the style of
        # rsl_rl / legged_gym may hand over bool dones or time_outs and float64 rewards: convert instead of misreading them.
        want = lambda t, dt: t if (t.dtype == dt and t.is_contiguous() and t.device == self.device) else t.to(device=self.device, dtype=dt).contiguous()
        rewards, dones = want(rewards.reshape(-1), torch.float32), want(dones.reshape(-1), torch.int64)
        time_outs = None if time_outs is None else want(time_outs.reshape(-1), torch.float32)
        if rewards.numel() != self.N or dones.numel() != self.N or (time_outs is not None and time_outs.numel() != self.N):
            raise ValueError(f"record: rewards / dones / time_outs must hold {self.N} entries")
        if self.update is not None and self.defer_record:
            self.flush()                                  # a still-pending record reads the PREVIOUS converted copies: launch it before they lose their last reference
        self._keep = (rewards, dones, time_outs)          # converted copies stay alive until the launch has read them
        eps, epi, n_ep, epa = (ep[0].data_ptr(), ep[1].data_ptr(), int(ep[1].numel()), ep[2].data_ptr()) if ep is not None else (None, None, 0, None)
        if self.update is not None and self.defer_record:
            # one-launch collection: nothing is launched here, the arguments ride at the head of the next act() (nm_ppo_record_act);
            # flush() files the last step of a rollout. The env's reward / done / time-out buffers stay valid until its next step().
            self._pending = (rewards.data_ptr(), dones.data_ptr(), None if time_outs is None else time_outs.data_ptr(), storage.values[s].data_ptr(),
                             float(gamma), storage.rewards[s].data_ptr(), storage.dones[s].data_ptr(), cur_ret.data_ptr(), cur_len.data_ptr(),
                             fin.data_ptr(), eps, epi, n_ep, epa)
            storage.step += 1
            return
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._L.nm_ppo_record(rewards.data_ptr(), dones.data_ptr(), None if time_outs is None else time_outs.data_ptr(),
                                         storage.values[s].data_ptr(), float(gamma), self.N, storage.rewards[s].data_ptr(),
                                         storage.dones[s].data_ptr(), cur_ret.data_ptr(), cur_len.data_ptr(), fin.data_ptr(), eps, epi, n_ep, epa, stream))
        storage.step += 1

    def can_rollout(self, env):
        """Can the whole rollout run as ONE launch inside the env's own waves (nm_rollout)? Needs this repo's fp32 env on the same device,
        networks of the compiled shape next to a FusedUpdate (its flat parameter vector is what the kernel reads), no privileged observations."""
        if self.update is None or not hasattr(env, "policy_rollout") or getattr(env, "_dtype", None) != _lib.DTYPE_F32:
            return False
        if env.num_envs != self.N or torch.device(env.device) != self.device or env.cfg.viewer.record_states or env.get_privileged_observations() is not None:
            return False
        a = (C.c_int32 * (len(self.a_lin) + 1))(self.a_lin[0].in_features, *[m.out_features for m in self.a_lin])
        c = (C.c_int32 * (len(self.c_lin) + 1))(self.c_lin[0].in_features, *[m.out_features for m in self.c_lin])
        return bool(self._L.nm_rollout_supported(a, c, len(self.a_lin)))

    def rollout(self, env, storage, steps, gamma, cur_ret, cur_len, fin, ep=None):
        """`steps` x (PPO.act, env.step, PPO.process_env_step + the runner's bookkeeping) as one launch; returns the last observation.
        Same noise keys as act(): (seed, iteration set by refresh(), step, env, action pair)."""
        self.drop_pending()
        storage.clear()
        if getattr(self, "_last_values", None) is None or self._last_values.numel() != env.num_envs:
            self._last_values = torch.zeros(env.num_envs, device=self.device)
        o = env.policy_rollout(steps, self.update.flat, self.seed, self.iter_dev, storage, gamma, cur_ret, cur_len, fin, ep=ep, last_values=self._last_values)
        self.last_values = self._last_values      # the value of the observation after the last step: PPO.compute_returns takes it (once)
        return o

    def drop_pending(self):
        """Forget a deferred record without launching it: the storage it would write into has been cleared (end of an update, a failed
        graph capture) - its arguments point at a step that no longer exists."""
        self._pending = None
        self.last_values = None

    def flush(self):
        """Launch the bookkeeping of a step whose act() successor has not come (the last step of a rollout). A no-op otherwise."""
        if self._pending is None:
            return
        rec, self._pending = self._pending, None
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._L.nm_ppo_record(rec[0], rec[1], rec[2], rec[3], rec[4], self.N, *rec[5:], stream))


class FusedUpdate:
    """PPO.update on the nm_ppo_* kernels. The module's parameters become views of ONE flat device vector (so state_dicts, the
    collector and everything else keep seeing live values), Adam's moments live in two more; the torch optimizer object is kept as the
    checkpoint container: its state entries are views of those vectors."""

    @staticmethod
    def supported(ac, device):
        if not FusedCollector.supported(ac, device):
            return False
        a = [m for m in ac.actor if isinstance(m, nn.Linear)]
        c = [m for m in ac.critic if isinstance(m, nn.Linear)]
        return all(x.out_features + y.out_features + 1 <= 128 for x, y in zip(a, c)) and a[0].in_features + 1 <= 128 and a[-1].out_features <= 32

    def __init__(self, ac, optimizer, device, lr):
        self.ac, self.opt, self.device = ac, optimizer, torch.device(device)
        self._L = _lib.load()
        a = [m for m in ac.actor if isinstance(m, nn.Linear)]
        c = [m for m in ac.critic if isinstance(m, nn.Linear)]
        self.params = [p for m in a for p in (m.weight, m.bias)] + [p for m in c for p in (m.weight, m.bias)] + [ac.std]
        assert len(self.params) == len(list(ac.parameters()))
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, device=self.device)
        self.m, self.v = torch.zeros(n, device=self.device), torch.zeros(n, device=self.device)
        off = 0
        self.views = []
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + k].view_as(p)
                self.views.append((off, k))
                off += k
        adims = (C.c_int32 * (len(a) + 1))(a[0].in_features, *[m.out_features for m in a])
        cdims = (C.c_int32 * (len(c) + 1))(c[0].in_features, *[m.out_features for m in c])
        h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        # The one-launch mini-batch step (k_ppo_step) synchronises its ~236 workgroups with a spinning grid barrier: they must all be
        # resident at once, i.e. the GPU must not be shared with another process's kernels. One rank per GPU (RCCL requires it) is fine;
        # the shared-card rehearsal of the multi-rank path over gloo is not - there the four-launch step is used (nm_ppo_create reads the
        # switch). A barrier that does time out makes nm_ppo_get_state fail loudly.
        import os
        import torch.distributed as dist
        shared = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and dist.get_backend() != "nccl"
        prev = os.environ.get("NM_PPO_UNFUSED_STEP")
        if shared:
            os.environ["NM_PPO_UNFUSED_STEP"] = "1"
        try:
            _lib.check(self._L.nm_ppo_create(adims, cdims, len(a), idx, C.byref(h)))
        finally:
            if shared:
                if prev is None:
                    os.environ.pop("NM_PPO_UNFUSED_STEP", None)
                else:
                    os.environ["NM_PPO_UNFUSED_STEP"] = prev
        self._h = h
        assert self._L.nm_ppo_num_params(h) == n
        self.A, self.n_obs = a[-1].out_features, a[0].in_features
        self.has_fast_path = bool(self._L.nm_ppo_has_fast_path(h))
        self.step_count = 0
        self._bind_optimizer_state()
        self.sync(lr)

    def _bind_optimizer_state(self):
        """Adam's per-parameter state as views of the flat moment vectors (what optimizer.state_dict() then saves)."""
        for p, (off, k) in zip(self.params, self.views):
            self.opt.state[p] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.m[off:off + k].view_as(p),
                                 "exp_avg_sq": self.v[off:off + k].view_as(p)}

    def sync(self, lr):
        """After the parameters (views of the flat vector) or the optimizer state were written by someone else - load_state_dict,
        a broadcast: re-read them."""
        for p, (off, k) in zip(self.params, self.views):      # a load may have re-pointed .data or replaced the optimizer's tensors
            if p.data.data_ptr() != self.flat[off:off + k].data_ptr():
                with torch.no_grad():
                    self.flat[off:off + k].copy_(p.detach().reshape(-1))
                    p.data = self.flat[off:off + k].view_as(p)
            st = self.opt.state.get(p)
            if st and "exp_avg" in st and st["exp_avg"].data_ptr() != self.m[off:off + k].data_ptr():
                self.m[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                self.step_count = int(float(st["step"]))
        self._bind_optimizer_state()
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._L.nm_ppo_sync_params(self._h, self.flat.data_ptr(), float(lr), int(self.step_count), stream))

    def minibatch(self, obs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, hp, phase=0, kl_override=-1.0, rows=None):
        """rows: int32 device tensor of row numbers - the mini-batch is obs[rows], actions[rows], ... gathered INSIDE the kernel
        (nm_ppo_minibatch_rows); None: the tensors are the mini-batch."""
        f = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()
        obs, actions, old_mu, old_sigma = f(obs), f(actions), f(old_mu), f(old_sigma)
        old_logp, advantages, returns, target_values = f(old_logp), f(advantages), f(returns), f(target_values)
        b1, b2 = self.opt.param_groups[0]["betas"]
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        if rows is not None and (rows.dtype != torch.int32 or not rows.is_contiguous() or rows.device != self.flat.device):
            raise ValueError("minibatch: rows must be a contiguous int32 tensor on the update's device")
        B = obs.shape[0] if rows is None else int(rows.numel())
        if rows is not None and getattr(self, "_storage_rows", None) != obs.shape[0]:
            _lib.check(self._L.nm_ppo_set_storage_rows(self._h, int(obs.shape[0])))      # 32-bit row offsets in the kernel: refused beyond 4 GiB per array
            self._storage_rows = obs.shape[0]
        _lib.check(self._L.nm_ppo_minibatch_rows(self._h, self.flat.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), obs.data_ptr(), actions.data_ptr(),
                                                 old_mu.data_ptr(), old_sigma.data_ptr(), old_logp.data_ptr(), advantages.data_ptr(), returns.data_ptr(),
                                                 target_values.data_ptr(), None if rows is None else rows.data_ptr(), B, obs.shape[1], hp["clip"], hp["value_coef"],
                                                 hp["entropy_coef"], int(hp["clip_value"]), hp["desired_kl"], int(hp["adaptive"]), hp["max_grad_norm"], b1, b2,
                                                 self.opt.param_groups[0]["eps"], phase, kl_override, stream))
        self._keep = (obs, actions, old_mu, old_sigma, old_logp, advantages, returns, target_values, rows)
        if phase != 1:
            self.step_count += 1

    def grad_and_kl(self, out=None):
        """[num_params + 1]: the last mini-batch's gradient in flat parameter order, then its mean KL (a copy on the device) - the
        vector a data-parallel update all-reduces."""
        out = torch.empty(self.flat.numel() + 1, device=self.device) if out is None else out
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._L.nm_ppo_copy_grad(self._h, out.data_ptr(), 0, stream))
        return out

    def grad(self):
        return self.grad_and_kl()[:-1]

    def set_grad_and_kl(self, g):
        assert g.numel() == self.flat.numel() + 1 and g.is_contiguous() and g.dtype == torch.float32
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._L.nm_ppo_copy_grad(self._h, g.data_ptr(), 1, stream))

    def permutation(self, n, seed, counter, out=None):
        """int32 [n]: a pseudo-random permutation of 0..n-1 keyed by (seed, counter), made on the device in one launch (nm_ppo_permutation)."""
        out = torch.empty(int(n), dtype=torch.int32, device=self.device) if out is None else out
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._L.nm_ppo_permutation(out.data_ptr(), int(n), int(seed) & 0xFFFFFFFFFFFFFFFF, int(counter), stream))
        return out

    def minibatch_data_parallel(self, *batch, hp, world, rows=None):
        """One mini-batch of a data-parallel update: local gradient (phase 1 writes gradient | KL straight into a buffer this object
        owns), ONE all-reduce of that buffer in place - averaged by RCCL itself when the process group is nccl, summed and divided on
        other backends -, then the step (phase 2 reads the buffer; every rank applies the identical step). Four launches per mini-batch
        (forward / backward, reduce, all-reduce, step), nothing read by the host: the whole update can be captured into a graph."""
        import torch.distributed as dist
        if getattr(self, "_gbuf", None) is None:
            self._gbuf = torch.zeros(self.flat.numel() + 1, device=self.device)
            _lib.check(self._L.nm_ppo_set_grad_buffer(self._h, C.c_void_p(self._gbuf.data_ptr())))
        self.minibatch(*batch, hp, phase=1, rows=rows)
        if dist.get_backend() == "nccl":
            dist.all_reduce(self._gbuf, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(self._gbuf)
            self._gbuf.div_(world)
        self.minibatch(*batch, hp, phase=2, rows=rows)

    def read_state(self, reset_sums=True):
        """dict(lr, steps, kl, value_loss_sum, surrogate_loss_sum, minibatches, clip_coef, grad_norm); one stream synchronisation."""
        out = (C.c_float * 8)()
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._L.nm_ppo_get_state(self._h, out, int(reset_sums), stream))
        keys = ("lr", "steps", "kl", "value_loss_sum", "surrogate_loss_sum", "minibatches", "clip_coef", "grad_norm")
        st = dict(zip(keys, [float(x) for x in out]))
        self.step_count = int(round(st["steps"]))         # the device's Adam step counter is the authority
        self.publish_step()
        return st

    STATE_KEYS = ("lr", "steps", "kl", "value_loss_sum", "surrogate_loss_sum", "minibatches", "clip_coef", "grad_norm")

    def snapshot_state(self, out9, reset_sums=True):
        """The values of read_state() copied into the DEVICE tensor out9 (9 floats; [8] != 0: the fused step's grid barrier timed out),
        stream-ordered and without a host synchronisation. The runner reads it one iteration later (state_from)."""
        assert out9.is_cuda and out9.dtype == torch.float32 and out9.numel() >= 9 and out9.is_contiguous()
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._L.nm_ppo_snapshot_state(self._h, C.c_void_p(out9.data_ptr()), int(reset_sums), stream))

    def state_from(self, vals9):
        """dict like read_state() from the nine host floats of a snapshot."""
        if vals9[8] != 0.0:
            raise _lib.NightmareHipError("nm_ppo: the grid barrier of the fused mini-batch step timed out (GPU shared with another job?); the steps since "
                                         "then were skipped (no parameter written); set NM_PPO_UNFUSED_STEP=1")
        st = dict(zip(self.STATE_KEYS, [float(x) for x in vals9[:8]]))
        # a snapshot is one iteration old when it is read: the host's own count (advanced by every mini-batch enqueued since) is never moved back
        self.step_count = max(self.step_count, int(round(st["steps"])))
        return st

    def publish_step(self):
        """Adam's step count into the torch optimizer's state (what a checkpoint saves next to the moments; rsl_rl saves
        optimizer.state_dict(), reference train.py:49-52 resumes from it)."""
        for p in self.params:
            st = self.opt.state.get(p)
            if st is not None:
                if torch.is_tensor(st.get("step")):
                    st["step"].fill_(float(self.step_count))
                else:
                    st["step"] = torch.tensor(float(self.step_count))

    def close(self):
        if getattr(self, "_h", None):
            self._L.nm_ppo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
