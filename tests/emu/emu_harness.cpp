// emu_harness.cpp — TEST-ONLY: runs the product kernel source (solo_step_kernel.h) on the CPU
// wave emulator.  Built by tests/emu/Makefile into libsolo_emu.so and driven from pytest.
#include "wave_emu.h"

#include "../../gym_solo_amd/csrc/solo_step_kernel.h"

#include <string>
#include <vector>

using namespace solo;

// steps > 1: one fused multi-step "launch" per robot (actions [steps][n][12], outputs [steps][n][.])
template <typename T>
static int run(const SoloConfig* cfg, const SoloModel* mdl, const SoloProgram* prog, int n,
               double* state, const double* snapshot, const double* actions, double* targets,
               const double* params, double* obs, double* reward, uint8_t* done,
               int32_t* term_count, double* stats, uint32_t flags, int steps = 1,
               const SoloTerrain* terrain = nullptr) {
  std::string err;
  if (int rc = validate_model(*mdl, &err)) { fprintf(stderr, "emu: %s\n", err.c_str()); return rc; }
  static KParams<T> P;
  pack_params<T>(*cfg, *mdl, &P);
  int D = 0;
  if (prog) {
    if (int rc = pack_program<T>(*prog, &P, &err)) { fprintf(stderr, "emu: %s\n", err.c_str()); return rc; }
    D = prog->num_obs;
  }
  auto conv = [](const double* src, size_t cnt) {
    std::vector<T> v(cnt);
    for (size_t i = 0; i < cnt; ++i) v[i] = (T)src[i];
    return v;
  };
  std::vector<T> st = conv(state, (size_t)n * SOLO_STATE_STRIDE);
  std::vector<T> snap = conv(snapshot, (size_t)n * SOLO_STATE_STRIDE);
  std::vector<T> tg = conv(targets, (size_t)n * SOLO_NUM_JOINTS);
  std::vector<T> act;
  if (actions) act = conv(actions, (size_t)steps * n * SOLO_NUM_JOINTS);
  std::vector<T> par = conv(params, (size_t)n * 4);
  std::vector<T> ob((size_t)steps * n * (D > 0 ? D : 1)), rew((size_t)steps * n);
  std::vector<T> terr;
  if (terrain) {
    terr = conv(terrain->heights, (size_t)terrain->nx * terrain->ny);
    P.terr_nx = terrain->nx; P.terr_ny = terrain->ny; P.terr_inv_cell = (T)(1.0 / terrain->cell);
    P.terr_ox = (T)terrain->origin[0]; P.terr_oy = (T)terrain->origin[1];
  }
  KBuffers<T> B;
  B.terrain = terrain ? terr.data() : nullptr;
  B.state = st.data(); B.snapshot = snap.data(); B.targets = tg.data();
  B.actions = actions ? act.data() : nullptr; B.params = par.data(); B.obs = ob.data();
  B.reward = rew.data(); B.done = done; B.term_count = term_count; B.stats = stats;
  B.num_envs = n; B.flags = flags; B.env_base = 0; B.steps = steps;
  B.action_stride = (long long)n * SOLO_NUM_JOINTS; B.obs_stride = (long long)n * D; B.reward_stride = n; B.done_stride = n;
  const KParams<T>* Pp = &P;
  for (int b = 0; b < n; ++b)
    WaveEmu::get().run_block(b, n, [&]() { solo_step_kernel<T, true>(Pp, B); });
  for (size_t i = 0; i < st.size(); ++i) state[i] = (double)st[i];
  for (size_t i = 0; i < tg.size(); ++i) targets[i] = (double)tg[i];
  if (flags & SOLO_STEP_OBS) for (size_t i = 0; i < (size_t)steps * n * D; ++i) obs[i] = (double)ob[i];
  if (flags & SOLO_STEP_REWARD) for (size_t i = 0; i < (size_t)steps * n; ++i) reward[i] = (double)rew[i];
  return 0;
}

extern "C" int solo_emu_step(const SoloConfig* cfg, const SoloModel* mdl, const SoloProgram* prog,
                             int dtype, int n, double* state, const double* snapshot,
                             const double* actions, double* targets, const double* params,
                             double* obs, double* reward, uint8_t* done, int32_t* term_count,
                             double* stats, uint32_t flags, const SoloTerrain* terrain) {
  if (dtype == SOLO_F32)
    return run<float>(cfg, mdl, prog, n, state, snapshot, actions, targets, params, obs, reward,
                      done, term_count, stats, flags, 1, terrain);
  return run<double>(cfg, mdl, prog, n, state, snapshot, actions, targets, params, obs, reward,
                     done, term_count, stats, flags, 1, terrain);
}

// fused multi-step launch: actions [steps][n][12]; obs [steps][n][D], reward [steps][n], done [steps][n]
extern "C" int solo_emu_rollout(const SoloConfig* cfg, const SoloModel* mdl, const SoloProgram* prog,
                                int dtype, int n, int steps, double* state, const double* snapshot,
                                const double* actions, double* targets, const double* params,
                                double* obs, double* reward, uint8_t* done, int32_t* term_count,
                                double* stats, uint32_t flags, const SoloTerrain* terrain) {
  if (dtype == SOLO_F32)
    return run<float>(cfg, mdl, prog, n, state, snapshot, actions, targets, params, obs, reward,
                      done, term_count, stats, flags, steps, terrain);
  return run<double>(cfg, mdl, prog, n, state, snapshot, actions, targets, params, obs, reward,
                     done, term_count, stats, flags, steps, terrain);
}
