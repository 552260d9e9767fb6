"""Per-stage wave time of the step kernel from in-kernel s_memtime stamps (needs the -DNM_STAMPS measurement build:
   make -C nightmare_rl_amd/csrc stamps ; NM_HIP_LIB=nightmare_rl_amd/csrc/libnightmare_hip_stamps.so python scripts/stamps.py)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = NightmareV3Config(); cfg.env.num_envs = N
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0     # 0.12 = the standing regime
acts = ((torch.rand(16, N, 18, generator=torch.Generator().manual_seed(0)) * 2 - 1) * scale).cuda()
env = NightmareV3Env(cfg, seed=0); env.reset()
for i in range(300 if scale == 1.0 else 1500): env.step(acts[i % 16])
out = (C.c_ulonglong * 16)()
env._L.nm_read_stamps(out, 1)
K = 200
for i in range(K): env.step(acts[i % 16])
env._L.nm_read_stamps(out, 0)
names = ["load", "check+normalise", "A smooth", "B floor collision", "B tibia pairs", "C rows+project (A matrix)", "C warm start+PGS",
         "C noslip", "C J'f+sensors", "D integrate", "epilogue: buffers + lane-0 stores", "epilogue: state stores", "epilogue: E3-E5 frames/termination", "epilogue: E6 reset",
         "epilogue: E7 rewards", "epilogue: E8 observation"]
if os.environ.get("NM_STAMPS_B"):   # measurement build with -DNM_STAMPS_B
    names[3] = "B (rest)"
    names[11:14] = ["B ring gather + support values", "B hill climbs / fallbacks", "B contact emission"]
    names[10] = "epilogue (all)"
    names[14], names[15] = "B   of which: hop ring gathers", "B   of which: exhaustive scan + ring"
waves = (N + 1) // 2
tot = sum(out[:16])
for k, n in enumerate(names):
    print(f"{n:28s} {out[k] / K / waves:10.0f} ticks/wave/step  {100.0 * out[k] / tot:5.1f} %")
c = env.counters()
print("fallbacks per wave-step:", c["hull_search_fallbacks"] / (env.common_step_counter * waves))
print(f"{'total':28s} {tot / K / waves:10.0f} ticks/wave/step (s_memtime ticks)")
