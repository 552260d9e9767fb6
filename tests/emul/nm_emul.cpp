// nm_emul.cpp - TEST SCAFFOLDING: compiles the device kernel source (nightmare_rl_amd/csrc/nm_core.h) for the
// host with -DNM_EMUL, where simt.h executes the 64 lanes of a wavefront in lockstep. Lets the exact device
// algorithm (lane mappings, LDS traffic, cross-lane reductions, solver sweeps) be checked against the CPU oracle
// in a container without a GPU. Never linked into the product library; the product has no CPU path.
#define NM_EMUL 1
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../nightmare_rl_amd/csrc/nm_host_model.h"

namespace {
struct EmuBase {
  virtual ~EmuBase() {}
  virtual void step(const float* actions, const double* cmd_u, float* obs, float* rew, int64_t* done, float* to, int nsub, int physics_only,
                    double* dbg, double* stat_sum, int* stat_cnt) = 0;
  virtual void get(int what, double* out) = 0;
  virtual void set(int what, const double* in) = 0;
  virtual int64_t* eplen() = 0;
  virtual void set_noise(const double* vec, const double* u) = 0;
  virtual void record(int env, double* out50) = 0;
  virtual void configure(const double* scales, const double* modes6) = 0;
  virtual void feet(int set, double* air, int* flags) = 0;
  virtual void hull_cache(int set, int* hc) = 0;
};

template <class real> struct Emu : EmuBase {
  int N;
  nmhost::Tables<real> T;
  nm::Model<real> M;
  std::vector<real> qpos, qvel, qwarm, dofpos, dofvel, act, cmd, epsum;
  std::vector<int64_t> ep;
  std::vector<uint32_t> ctr;
  std::vector<int> hcache, feetflags;
  std::vector<real> feetair;
  uint64_t seed;
  int64_t off;
  int G = 1;
  std::vector<real> nvec, nu, rec = std::vector<real>(64, 0);
  uint64_t nstep = 0;
  int rec_env = -1;
  void set_noise(const double* vec, const double* u) override {
    nvec.clear(); nu.clear();
    if (vec) nvec.assign(vec, vec + 66);
    if (u) nu.assign(u, u + (size_t)N * 66);
  }
  void record(int env, double* out50) override {
    rec_env = env;
    if (out50) for (int i = 0; i < 50; i++) out50[i] = (double)rec[i];
  }
  void configure(const double* scales, const double* m) override {   // raw reward scales [kNREW] + (tibia mode, max, body mode, max, base height target, max contact force)
    nmhost::EnvConfig cfg;
    for (int k = 0; k < nm::kNREW; k++) cfg.rew_scales[k] = scales[k];
    cfg.tibia_contact_mode = (int)m[0]; cfg.tibia_max_contact_force = m[1]; cfg.body_contact_mode = (int)m[2]; cfg.body_max_contact_force = m[3];
    cfg.base_height_target = m[4]; cfg.max_contact_force = m[5];
    T.fill_scalars(M, cfg);
  }
  void hull_cache(int set, int* hc) override {
    for (size_t i = 0; i < hcache.size(); i++) { if (set) hcache[i] = hc[i]; else hc[i] = hcache[i]; }
  }
  void feet(int set, double* air, int* flags) override {
    for (size_t i = 0; i < feetair.size(); i++) { if (set) feetair[i] = (real)air[i]; else air[i] = (double)feetair[i]; }
    for (int i = 0; i < N; i++) { if (set) feetflags[i] = flags[i]; else flags[i] = feetflags[i]; }
  }
  Emu(int n, uint64_t s, int64_t o, int g) : N(n), seed(s), off(o), G(g) {
    T.build();
    nmhost::EnvConfig cfg;
    T.fill_scalars(M, cfg);
    feetair.assign((size_t)N * 6, 0); feetflags.assign(N, 0);
    M.hullv = T.hullv.data(); M.hullnv = T.hullnv.data();
    qpos.assign((size_t)N * 25, 0); qvel.assign((size_t)N * 24, 0); qwarm.assign((size_t)N * 24, 0);
    dofpos.assign((size_t)N * 18, 0); dofvel.assign((size_t)N * 18, 0); act.assign((size_t)N * 18, 0);
    cmd.assign((size_t)N * 3, 0); epsum.assign((size_t)N * nm::kNREW, 0); ep.assign(N, 0); ctr.assign(N, 0); hcache.assign((size_t)N * 8, 0);
    for (int i = 0; i < N; i++)
      for (int j = 0; j < 25; j++) qpos[(size_t)i * 25 + j] = T.qpos0[j];
  }
  std::vector<real>* arr(int what) {
    switch (what) {
      case 0: return &qpos; case 1: return &qvel; case 2: return &qwarm; case 3: return &dofpos; case 4: return &dofvel;
      case 5: return &act; case 6: return &cmd; case 7: return &epsum;
    }
    return nullptr;
  }
  void get(int what, double* out) override { auto* a = arr(what); for (size_t i = 0; i < a->size(); i++) out[i] = (double)(*a)[i]; }
  void set(int what, const double* in) override { auto* a = arr(what); for (size_t i = 0; i < a->size(); i++) (*a)[i] = (real)in[i]; }
  int64_t* eplen() override { return ep.data(); }
  void step(const float* actions, const double* cmd_u, float* obs, float* rew, int64_t* done, float* to, int nsub, int physics_only,
            double* dbg, double* stat_sum, int* stat_cnt) override {
    std::vector<real> cu, dbgr((size_t)N * nm::kDbgN, 0), ssum(nm::kNREW, 0);
    if (cmd_u) { cu.resize((size_t)N * 4); for (size_t i = 0; i < cu.size(); i++) cu[i] = (real)cmd_u[i]; }
    int scnt[4] = {0, 0, 0, 0};
    nm::Args<real> A{};
    A.N = N; A.seed = seed; A.env_offset = off;
    A.qpos = qpos.data(); A.qvel = qvel.data(); A.qwarm = qwarm.data(); A.dofpos = dofpos.data(); A.dofvel = dofvel.data();
    A.act = act.data(); A.cmd = cmd.data(); A.epsum = epsum.data(); A.feetair = feetair.data(); A.feetflags = feetflags.data(); A.eplen = ep.data(); A.rngctr = ctr.data(); A.hullcache = hcache.data();
    A.actions = actions; A.cmd_u = cmd_u ? cu.data() : nullptr;
    A.obs = obs; A.rew = rew; A.timeout_now = to; A.done = done; A.stat_sum = ssum.data(); A.stat_cnt = scnt;
    A.dbg = dbg ? dbgr.data() : nullptr; A.nsub = nsub; A.physics_only = physics_only;
    if (!physics_only) {
      A.noise_vec = nvec.empty() ? nullptr : nvec.data(); A.noise_u = nu.empty() ? nullptr : nu.data(); A.noise_step = nstep++;
      A.rec = rec_env >= 0 ? rec.data() : nullptr; A.rec_env = rec_env;
    }
    if (G == 1) {
      static thread_local nm::ShW<real, 1> sh;
      for (int wv = 0; wv < N; wv++) nm::wave_step<real, 1>(sh, M, A, wv);
    } else {
      static thread_local nm::ShW<real, 2> sh;
      for (int wv = 0; wv * 2 < N; wv++) nm::wave_step<real, 2>(sh, M, A, wv);
    }
    if (dbg) for (size_t i = 0; i < dbgr.size(); i++) dbg[i] = (double)dbgr[i];
    if (stat_sum) for (int k = 0; k < nm::kNREW; k++) stat_sum[k] = (double)ssum[k];
    if (stat_cnt) { stat_cnt[0] = scnt[0]; stat_cnt[1] = scnt[1]; }
  }
};
}  // namespace

extern "C" {
void* emu_create(int N, int use_double, uint64_t seed, int64_t env_off, int envs_per_wave) {
  if (use_double) return new Emu<double>(N, seed, env_off, envs_per_wave);
  return new Emu<float>(N, seed, env_off, envs_per_wave);
}
void emu_destroy(void* h) { delete (EmuBase*)h; }
void emu_step(void* h, const float* actions, const double* cmd_u, float* obs, float* rew, int64_t* done, float* to, int nsub, int physics_only,
              double* dbg, double* stat_sum, int* stat_cnt) {
  ((EmuBase*)h)->step(actions, cmd_u, obs, rew, done, to, nsub, physics_only, dbg, stat_sum, stat_cnt);
}
void emu_get(void* h, int what, double* out) { ((EmuBase*)h)->get(what, out); }
void emu_set(void* h, int what, const double* in) { ((EmuBase*)h)->set(what, in); }
int64_t* emu_eplen(void* h) { return ((EmuBase*)h)->eplen(); }
void emu_set_noise(void* h, const double* vec, const double* u) { ((EmuBase*)h)->set_noise(vec, u); }
void emu_record(void* h, int env, double* out50) { ((EmuBase*)h)->record(env, out50); }
void emu_configure(void* h, const double* scales, const double* modes6) { ((EmuBase*)h)->configure(scales, modes6); }
void emu_feet(void* h, int set, double* air, int* flags) { ((EmuBase*)h)->feet(set, air, flags); }
void emu_hull_cache(void* h, int set, int* hc) { ((EmuBase*)h)->hull_cache(set, hc); }
int emu_nrew() { return nm::kNREW; }
int emu_dbg_n() { return nm::kDbgN; }
}
extern "C" long emu_together_count() { return nm::nm_emul_together(); }
