import os, sys, torch, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.helpers import class_to_dict
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config, NightmareV3ConfigPPO
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
from nightmare_rl_amd.rl import OnPolicyRunner
cfg = NightmareV3Config(); cfg.env.num_envs = 256
env = NightmareV3Env(cfg, seed=0)
tc = class_to_dict(NightmareV3ConfigPPO()); tc["runner"]["num_steps_per_env"] = 4
r = OnPolicyRunner(env, tc, log_dir=None, device="cuda:0")
alg = r.alg
obs = env.get_observations()
def attempt(name, fn):
    torch.cuda.synchronize()
    try:
        with torch.inference_mode():
            fn()                      # warm
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            g.replay(); torch.cuda.synchronize()
        print("OK  ", name, flush=True)
    except Exception as e:
        print("FAIL", name, type(e).__name__, str(e).split("\n")[0], flush=True)
        try: torch.cuda.synchronize()
        except Exception: pass
cur = torch.zeros(256, device="cuda")
attempt("actor forward", lambda: alg.actor_critic.actor(obs))
attempt("normal sample", lambda: alg.actor_critic.act(obs))
m = torch.zeros(256, 18, device="cuda")
attempt("randn_like", lambda: m + 0.5 * torch.randn_like(m))
attempt("torch.normal(t,t)", lambda: torch.normal(m, m + 1.0))
attempt("normal_() in place", lambda: torch.empty_like(m).normal_())
from torch.distributions import Normal
attempt("Normal ctor", lambda: Normal(m, m * 0.0 + 1.0))
attempt("Normal log_prob", lambda: Normal(m, m * 0.0 + 1.0).log_prob(m).sum(-1))
attempt("alg.act", lambda: alg.act(obs, obs))
acts = torch.zeros(256, 18, device="cuda")
attempt("env.step", lambda: env.step(acts))
def pes():
    a = alg.act(obs, obs)
    o, _, rew, done, infos = env.step(a)
    alg.storage.step = 0
    alg.process_env_step(rew, done, infos)
attempt("act+step+process", pes)
def stats():
    d = (env.reset_buf > 0).float()
    cur.add_(env.rew_buf); cur.mul_(1 - d)
    e = torch.stack([env.extras["episode"][k].float() for k in sorted(env.extras["episode"])])
    return e.sum()
attempt("stats", stats)
