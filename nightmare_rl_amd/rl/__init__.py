"""On-policy RL stack with the surface of rsl_rl v1.0.2 (the reference's trainer: train.py:1,40,54), kept on the GPU."""
from .actor_critic import ActorCritic
from .ppo import PPO
from .storage import RolloutStorage
from .runner import OnPolicyRunner

__all__ = ["ActorCritic", "PPO", "RolloutStorage", "OnPolicyRunner"]
