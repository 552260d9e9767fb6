"""ActorCritic: separate actor / critic MLPs + a learned, state-independent action std (rsl_rl v1.0.2
`modules/actor_critic.py` semantics; reference configuration envs/nightmare_v3_config.py:105-109). Parameter names
(`actor.<i>.weight`, `critic.<i>.weight`, `std`) match upstream so `model_<it>.pt['model_state_dict']` checkpoints
(reference play.py:71) load in both directions."""
import torch
import torch.nn as nn
from torch.distributions import Normal

_ACTIVATIONS = {"elu": nn.ELU, "selu": nn.SELU, "relu": nn.ReLU, "lrelu": nn.LeakyReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid}


class _SplitKLinearFn(torch.autograd.Function):
    """y = x W' + b whose weight gradient dW = dY' X is computed as a batched product over row chunks and then summed.
    A PPO mini-batch has ~80 k rows and these layers are <= 66 wide: as one GEMM (K = 80 k, M x N tiny) the library runs a
    handful of workgroups for 250 us; 128 chunks of 640 rows keep the whole GPU busy for a few microseconds."""

    @staticmethod
    def forward(ctx, x, w, b, chunks):
        ctx.save_for_backward(x, w)
        ctx.chunks = chunks
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        s = ctx.chunks
        gx = gy @ w if ctx.needs_input_grad[0] else None
        rows = x.shape[0] // s
        gw = torch.bmm(gy.reshape(s, rows, gy.shape[1]).transpose(1, 2), x.reshape(s, rows, x.shape[1])).sum(dim=0)
        return gx, gw, gy.sum(dim=0), None


class SplitKLinear(nn.Linear):
    """nn.Linear (same parameters, same state_dict keys) with the split-K weight gradient for tall inputs."""
    CHUNKS, MIN_ROWS = 128, 16384

    def forward(self, x):
        if torch.is_grad_enabled() and x.dim() == 2 and x.shape[0] >= self.MIN_ROWS and x.shape[0] % self.CHUNKS == 0 and self.weight.requires_grad:
            return _SplitKLinearFn.apply(x, self.weight, self.bias, self.CHUNKS)
        return super().forward(x)


def _mlp(n_in, hidden, n_out, act):
    dims = [n_in] + list(hidden)
    layers = []
    for a, b in zip(dims[:-1], dims[1:]):
        layers += [SplitKLinear(a, b), act()]
    layers.append(SplitKLinear(dims[-1], n_out))
    return nn.Sequential(*layers)


class ActorCritic(nn.Module):
    is_recurrent = False

    def __init__(self, num_actor_obs, num_critic_obs, num_actions, actor_hidden_dims=(256, 256, 256), critic_hidden_dims=(256, 256, 256),
                 activation="elu", init_noise_std=1.0, **kwargs):
        super().__init__()
        if kwargs:
            print("ActorCritic: ignoring unexpected arguments " + str(list(kwargs)))
        act = _ACTIVATIONS[activation]
        self.actor = _mlp(num_actor_obs, actor_hidden_dims, num_actions, act)
        self.critic = _mlp(num_critic_obs, critic_hidden_dims, 1, act)
        self.std = nn.Parameter(init_noise_std * torch.ones(num_actions))
        self.distribution = None
        self.actor_dims = [num_actor_obs] + list(actor_hidden_dims) + [num_actions]
        self.activation_name = activation
        Normal.set_default_validate_args(False)

    def reset(self, dones=None):
        pass

    def forward(self):
        raise NotImplementedError

    @property
    def action_mean(self):
        return self.distribution.mean

    @property
    def action_std(self):
        return self.distribution.stddev

    @property
    def entropy(self):
        return self.distribution.entropy().sum(dim=-1)

    def update_distribution(self, observations):
        mean = self.actor(observations)
        self.distribution = Normal(mean, mean * 0.0 + self.std)

    def act(self, observations, **kwargs):
        self.update_distribution(observations)
        # mean + std * N(0,1): the same draw as Normal.sample(), written with randn_like because torch.normal(tensor, tensor)
        # cannot be captured into a HIP graph on this stack (the runner replays the whole rollout as one graph)
        d = self.distribution
        return (d.mean + d.stddev * torch.randn_like(d.mean)).detach()

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(dim=-1)

    def act_inference(self, observations):
        return self.actor(observations)

    def evaluate(self, critic_observations, **kwargs):
        return self.critic(critic_observations)

    def actor_linears(self):
        return [m for m in self.actor if isinstance(m, nn.Linear)]
