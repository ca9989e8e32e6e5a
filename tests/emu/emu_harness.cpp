// emu_harness.cpp — TEST-ONLY: runs the product kernel source (solo_step_kernel.h) on the CPU
// wave emulator.  Built by tests/emu/Makefile into libsolo_emu.so and driven from pytest.
#include "wave_emu.h"

#ifdef SOLO_EMU_TRACE
// DIAGNOSTIC variant (make trace): records every lane's impulse / candidate after each Gauss-Seidel
// sweep of the LAST emulated robot-step, for tools/analyse_slow_steps.py
static double g_trace_lam[64][64], g_trace_v[64][64];
static unsigned long long g_trace_pend[64];
static int g_trace_sweeps = 0;
#define SOLO_PGS_SWEEP_HOOK(it, pend, lam, v)                                       \
  do {                                                                              \
    if ((it) < 64) {                                                                \
      g_trace_lam[(it)][solo::lane_id()] = (double)(lam);                           \
      g_trace_v[(it)][solo::lane_id()] = (double)(v);                               \
      g_trace_pend[(it)] = (pend);                                                  \
      g_trace_sweeps = (it) + 1;                                                    \
    }                                                                               \
  } while (0)
extern "C" int solo_emu_trace(double* lam, double* v, unsigned long long* pend) {
  for (int i = 0; i < 64 * 64; ++i) { lam[i] = (&g_trace_lam[0][0])[i]; v[i] = (&g_trace_v[0][0])[i]; }
  for (int i = 0; i < 64; ++i) pend[i] = g_trace_pend[i];
  return g_trace_sweeps;
}
#endif

#include "../../gym_solo_amd/csrc/solo_step_kernel.h"

#include <string>
#include <vector>

using namespace solo;

// Gauss-Seidel sweeps each robot ran in the last emulated launch (view.cost of the engine)
static std::vector<int32_t> g_last_cost;
static int32_t g_fault = 0;      // the engine's fault word (KBuffers::fault), as the emulator sees it
static int g_sabotage = 0;
extern "C" int solo_emu_last_cost(int32_t* out, int n) {
  const int m = (int)g_last_cost.size() < n ? (int)g_last_cost.size() : n;
  for (int i = 0; i < m; ++i) out[i] = g_last_cost[i];
  return m;
}

// the step kernel's workgroup -> robot map (solo_kernel_params.h), for tests/test_emu_kernel.py
extern "C" int solo_emu_xcd_contiguous(int b, int count) { return solo::xcd_contiguous(b, count); }

// steps > 1: one fused multi-step "launch" per robot (actions [steps][n][12], outputs [steps][n][.])
template <typename T>
static int run(const SoloConfig* cfg, const SoloModel* mdl, const SoloProgram* prog, int n,
               double* state, const double* snapshot, const double* actions, double* targets,
               const double* params, double* obs, double* reward, uint8_t* done,
               int32_t* term_count, double* stats, uint32_t flags, int steps = 1,
               const SoloTerrain* terrain = nullptr, double* warm = nullptr) {
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
  std::vector<T> wrm;   // the warm-start cache (SoloConfig::solver_warm_start), as Engine::buffers passes it
  if (warm != nullptr && cfg->solver_warm_start > 0) wrm = conv(warm, (size_t)n * 64);
  std::vector<T> ob((size_t)steps * n * (D > 0 ? D : 1)), rew((size_t)steps * n);
  std::vector<T> traj((size_t)steps * n * SOLO_STATE_STRIDE);
  std::vector<T> terr;
  if (terrain) {
    terr = conv(terrain->heights, (size_t)terrain->nx * terrain->ny);
    P.c.terr_nx = terrain->nx; P.c.terr_ny = terrain->ny; P.c.terr_inv_cell = (T)(1.0 / terrain->cell);
    P.c.terr_ox = (T)terrain->origin[0]; P.c.terr_oy = (T)terrain->origin[1];
  }
  const bool want_obs = (flags & SOLO_STEP_OBS) != 0, want_reward = (flags & SOLO_STEP_REWARD) != 0;
  KBuffers<T> B;
  B.terrain = terrain ? terr.data() : nullptr;
  g_last_cost.assign((size_t)n, 0);
  B.order = nullptr; B.cost = g_last_cost.data();
  B.state = st.data(); B.snapshot = snap.data(); B.targets = tg.data();
  B.actions = actions ? act.data() : nullptr; B.params = par.data();
  // as Engine::launch_chain: a single-step f32 launch evaluates its outputs lane-parallel over the items of the
  // step; every other launch leaves records and evaluates them in its output epilogue (lane = step), recording
  // every step into the caller's [steps][n][.] buffers
  const bool inline_outputs = steps == 1 && (want_obs || want_reward) && kInlineOutputs<T, true>;
  B.traj = (!inline_outputs && (want_obs || want_reward)) ? traj.data() : nullptr;
  B.obs_inline = (inline_outputs && want_obs) ? ob.data() : nullptr;
  B.reward_inline = (inline_outputs && want_reward) ? rew.data() : nullptr;
  B.obs_rec = (!inline_outputs && want_obs) ? ob.data() : nullptr;
  B.reward_rec = (!inline_outputs && want_reward) ? rew.data() : nullptr;
  B.obs_rec_stride = (long long)n * D; B.reward_rec_stride = n; B.obs_from = 0;
  B.view_obs = B.view_reward = nullptr; B.view_done = nullptr;
  B.done = done; B.term_count = term_count; B.stats = stats;
  B.num_envs = n; B.flags = flags; B.env_base = 0; B.count = n; B.steps = steps;
  B.action_stride = (long long)n * SOLO_NUM_JOINTS; B.done_stride = n;
  // robot migration (SoloConfig::migrate_steps), as Engine::launch_chain sets it up: chunks of the launch's steps go
  // through the queue; the emulated waves run one after the other, so the first drains every ring it can reach
  std::vector<int32_t> queue;
  B.queue = nullptr; B.q_rings = 1; B.q_chunk = 0; B.fault = &g_fault;
  B.warm = wrm.empty() ? nullptr : wrm.data();
  if (cfg->migrate_steps > 0 && steps > cfg->migrate_steps && (flags & SOLO_STEP_PHYSICS) && flags != SOLO_STEP_PHYSICS) {  // (as the engine: stepSimulation-only launches do not migrate)
    B.q_chunk = migration_chunk_steps(steps, cfg->migrate_steps);
    B.q_rings = n == 16 ? 8 : migration_rings(n);  // (16 robots: eight rings of two, so that the CPU suite walks several rings too)
    queue.resize(migration_queue_ints(n, steps, B.q_chunk) + 1);
    for (size_t i = 0; i + 1 < queue.size(); ++i) migration_queue_init(queue.data(), i, 0, n, B.q_rings, steps, B.q_chunk, nullptr);
    queue.back() = -1;
    // FAULT INJECTION (tests/test_emu_kernel.py): ring 0's tail starts one slot too far - its first chunk-1 slot is never
    // published, the wave that holds that slot's ticket must give up (bounded wait), count itself and set the fault word
    if (g_sabotage) queue[16] += 1;
    B.queue = queue.data();
  }
  const KParams<T>* Pp = &P;
  // the step kernel, output epilogue included: one emulated wavefront per robot (as Engine::launch_chain launches it)
  for (int b = 0; b < n; ++b)
    WaveEmu::get().run_block(b, n, [&]() {
      // (as Engine::launch_chain: pybullet's residual threshold is a kernel instantiation of its own)
      if (B.queue != nullptr) {
        if (cfg->solver_residual_threshold > 0) solo_step_kernel<T, true, true, true>(Pp, B);
        else solo_step_kernel<T, true, false, true>(Pp, B);
      } else if (cfg->solver_residual_threshold > 0) solo_step_kernel<T, true, true>(Pp, B);
      else solo_step_kernel<T, true, false>(Pp, B);
    });
  for (size_t i = 0; i < st.size(); ++i) state[i] = (double)st[i];
  for (size_t i = 0; i < tg.size(); ++i) targets[i] = (double)tg[i];
  for (size_t i = 0; i < wrm.size(); ++i) warm[i] = (double)wrm[i];
  if (flags & SOLO_STEP_OBS) for (size_t i = 0; i < (size_t)steps * n * D; ++i) obs[i] = (double)ob[i];
  if (flags & SOLO_STEP_REWARD) for (size_t i = 0; i < (size_t)steps * n; ++i) reward[i] = (double)rew[i];
  return 0;
}

extern "C" int solo_emu_step(const SoloConfig* cfg, const SoloModel* mdl, const SoloProgram* prog,
                             int dtype, int n, double* state, const double* snapshot,
                             const double* actions, double* targets, const double* params,
                             double* obs, double* reward, uint8_t* done, int32_t* term_count,
                             double* stats, uint32_t flags, const SoloTerrain* terrain, double* warm) {
  if (dtype == SOLO_F32)
    return run<float>(cfg, mdl, prog, n, state, snapshot, actions, targets, params, obs, reward,
                      done, term_count, stats, flags, 1, terrain, warm);
  return run<double>(cfg, mdl, prog, n, state, snapshot, actions, targets, params, obs, reward,
                     done, term_count, stats, flags, 1, terrain, warm);
}

// fused multi-step launch: actions [steps][n][12]; obs [steps][n][D], reward [steps][n], done [steps][n]
extern "C" int solo_emu_rollout(const SoloConfig* cfg, const SoloModel* mdl, const SoloProgram* prog,
                                int dtype, int n, int steps, double* state, const double* snapshot,
                                const double* actions, double* targets, const double* params,
                                double* obs, double* reward, uint8_t* done, int32_t* term_count,
                                double* stats, uint32_t flags, const SoloTerrain* terrain, double* warm) {
  if (dtype == SOLO_F32)
    return run<float>(cfg, mdl, prog, n, state, snapshot, actions, targets, params, obs, reward,
                      done, term_count, stats, flags, steps, terrain, warm);
  return run<double>(cfg, mdl, prog, n, state, snapshot, actions, targets, params, obs, reward,
                     done, term_count, stats, flags, steps, terrain, warm);
}

// the fault word a wave sets when it gives up waiting for a ring slot (and clears it); fault injection on / off
extern "C" int solo_emu_take_fault(void) { const int f = g_fault; g_fault = 0; return f; }
extern "C" void solo_emu_sabotage_queue(int on) { g_sabotage = on; }
