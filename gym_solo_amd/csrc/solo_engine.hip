// solo_engine.hip — implementation of the C-ABI in include/solo_engine.h for gfx950.
// Host logic only: buffer ownership, parameter upload, launches.  The arithmetic is in
// solo_step_kernel.h.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "solo_wave_ops.h"
#include "solo_pgs_gfx950.h"  // (defines SOLO_PGS_GFX950: the f32 Gauss-Seidel loop of the step kernel in assembly)
#include "solo_step_kernel.h"

// TRANSLATION UNITS (round 4).  The product library is this file compiled TWICE (Makefile): -DSOLO_TU_F32 = the C ABI and
// the f32 engine, -DSOLO_TU_F64 = the f64 engine alone, with `-mllvm -disable-machine-licm`: the f64 step kernel lives on
// exactly 168 VGPRs (three waves per SIMD), and what machine LICM hoists out of its step loop (LDS base addresses, flags)
// it then has to SPILL - scratch reloads inside the step, each behind an s_waitcnt vmcnt(0) that also waits for the
// step's freshly issued action load.  Without the pass: 0 VGPR spills, +3 ... 5 % (profiles/round4_ab.log); the f32
// kernels (0 spills either way) are 1 ... 3 % faster WITH it, hence two units.  Without either macro (the test and
// diagnostic builds) everything is one unit, as before.
namespace solo_engine_detail {

struct EngineBase {
  virtual ~EngineBase() {}
  virtual int set_program(const SoloProgram* p) = 0;
  virtual int reset(const uint8_t* mask, hipStream_t s) = 0;
  virtual int settle(hipStream_t s) = 0;
  virtual int set_targets(const void* a, hipStream_t s) = 0;
  virtual int step(const void* a, uint32_t flags, hipStream_t s) = 0;
  virtual int rollout(const void* a, int k, uint32_t flags, void* obs_out, void* reward_out, void* done_out, hipStream_t s) = 0;
  virtual int view(SoloStateView* v) = 0;
  virtual int set_params(int which, const void* p, hipStream_t s) = 0;
  virtual int set_terrain(const SoloTerrain* t, hipStream_t s) = 0;
  virtual int set_order(const int32_t* order, hipStream_t s) = 0;
  virtual int time_step(const void* a, uint32_t flags, int reps, hipStream_t s, double* ms) = 0;
  virtual int time_rollout(const void* a, int k, uint32_t flags, void* obs_out, void* reward_out, void* done_out, hipStream_t s, double* ms) = 0;
  virtual int plan(int k, SoloLaunchPlan* out) = 0;
  virtual int reserve(int k, uint32_t flags) = 0;
  virtual int check_fault() = 0;
  virtual const char* kernel_name() = 0;
  std::string err;
};

#define HIP_TRY(expr)                                                                 \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      err = std::string(#expr) + ": " + hipGetErrorString(e_);                        \
      return SOLO_ERR_HIP;                                                            \
    }                                                                                 \
  } while (0)

constexpr size_t kStatsBytes = (size_t)SOLO_STATS_SHARDS * SOLO_STATS_WIDTH * sizeof(double);

template <typename T>
struct Engine final : EngineBase {
  SoloConfig cfg;
  SoloModel model;
  int n = 0, device = 0, obs_dim = 0;
  bool have_program = false;
  solo::KParams<T> hparams;
  solo::KParams<T>* dparams = nullptr;
  T *state = nullptr, *snapshot = nullptr, *targets = nullptr, *params = nullptr, *obs = nullptr,
    *reward = nullptr, *settle_actions = nullptr, *warm = nullptr;  // (warm: the warm-start cache [N][64], SoloStateView::warm)
  // per-launch scratch: the step records [N][S][32] a fused launch leaves for its own output epilogue
  // (S = steps per launch)
  T* traj = nullptr;
  uint8_t* done = nullptr;
  int32_t* term_count = nullptr;
  int32_t *order = nullptr, *cost = nullptr;  // launch order (null = identity), per-robot sweeps of the last launch
  bool use_order = false;
  double* stats = nullptr;
  T* terrain = nullptr;
  int32_t* queue = nullptr;   // robot-migration queues of the launches in flight (one region per rollout slice), or null
  size_t queue_ints = 0;      // (allocated lazily, for the geometry of the rollout at hand)
  int traj_steps = 0;         // steps the record scratch `traj` holds per robot (allocated lazily)
  int32_t* fault_host = nullptr;  // pinned host word a wave that gives up waiting sets (SOLO_ERR_INCOMPLETE), device-visible
  int32_t* fault_dev = nullptr;
#ifdef SOLO_STAMPS
  unsigned long long* stamps = nullptr;
#endif

  ~Engine() override {
    (void)hipSetDevice(device);
    for (int g = 0; g < kMaxStreams; ++g) {
      if (sub[g]) (void)hipStreamDestroy(sub[g]);
      if (ev_join[g]) (void)hipEventDestroy(ev_join[g]);
    }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    for (void* p : {(void*)dparams, (void*)state, (void*)snapshot, (void*)targets, (void*)params,
                    (void*)obs, (void*)reward, (void*)settle_actions, (void*)done,
                    (void*)term_count, (void*)order, (void*)cost, (void*)stats, (void*)terrain, (void*)traj, (void*)queue, (void*)warm})
      if (p) (void)hipFree(p);
    if (fault_host) (void)hipHostFree(fault_host);
  }

  int init(const SoloConfig& c, const SoloModel& m, int num_envs, int dev) {
    cfg = c; model = m; n = num_envs; device = dev;
    HIP_TRY(hipSetDevice(device));
    solo::pack_params<T>(cfg, model, &hparams);
    const size_t ns = (size_t)n * SOLO_STATE_STRIDE;
    HIP_TRY(hipMalloc((void**)&dparams, sizeof(hparams)));
    HIP_TRY(hipMalloc((void**)&state, ns * sizeof(T)));
    HIP_TRY(hipMalloc((void**)&snapshot, ns * sizeof(T)));
    HIP_TRY(hipMalloc((void**)&targets, (size_t)n * SOLO_NUM_JOINTS * sizeof(T)));
    HIP_TRY(hipMalloc((void**)&settle_actions, (size_t)n * SOLO_NUM_JOINTS * sizeof(T)));
    HIP_TRY(hipMalloc((void**)&params, (size_t)n * 4 * sizeof(T)));
    HIP_TRY(hipMalloc((void**)&obs, (size_t)n * SOLO_MAX_OBS * sizeof(T)));
    HIP_TRY(hipMalloc((void**)&reward, (size_t)n * sizeof(T)));
    HIP_TRY(hipMalloc((void**)&done, (size_t)n));
    HIP_TRY(hipMalloc((void**)&term_count, (size_t)n * SOLO_MAX_TERMS * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void**)&stats, kStatsBytes));
    HIP_TRY(hipMalloc((void**)&order, (size_t)n * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void**)&cost, (size_t)n * sizeof(int32_t)));
    HIP_TRY(hipMemset(cost, 0, (size_t)n * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void**)&warm, (size_t)n * 64 * sizeof(T)));
    HIP_TRY(hipMemset(warm, 0, (size_t)n * 64 * sizeof(T)));
    HIP_TRY(hipHostMalloc((void**)&fault_host, sizeof(int32_t), hipHostMallocMapped));
    *fault_host = 0;
    HIP_TRY(hipHostGetDevicePointer((void**)&fault_dev, fault_host, 0));
    HIP_TRY(hipMemset(obs, 0, (size_t)n * SOLO_MAX_OBS * sizeof(T)));
    HIP_TRY(hipMemset(reward, 0, (size_t)n * sizeof(T)));
    HIP_TRY(hipMemset(done, 0, (size_t)n));
    HIP_TRY(hipMemset(term_count, 0, (size_t)n * SOLO_MAX_TERMS * sizeof(int32_t)));
    HIP_TRY(hipMemset(stats, 0, kStatsBytes));
#ifdef SOLO_STAMPS
    HIP_TRY(hipMalloc((void**)&stamps, (size_t)n * 32 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(stamps, 0, (size_t)n * 32 * sizeof(unsigned long long)));
#endif
    std::vector<T> hp((size_t)n * 4, T(0));
    std::vector<T> ha((size_t)n * SOLO_NUM_JOINTS);
    for (int e = 0; e < n; ++e) {
      hp[(size_t)e * 4 + 0] = (T)cfg.lateral_friction;
      hp[(size_t)e * 4 + 1] = T(1);
      // settle targets are absolute radians; the kernel multiplies actions by action_scale
      for (int j = 0; j < SOLO_NUM_JOINTS; ++j)
        ha[(size_t)e * SOLO_NUM_JOINTS + j] = (T)(cfg.settle_targets[j] / cfg.action_scale);
    }
    HIP_TRY(hipMemcpy(params, hp.data(), hp.size() * sizeof(T), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(settle_actions, ha.data(), ha.size() * sizeof(T), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dparams, &hparams, sizeof(hparams), hipMemcpyHostToDevice));
    return settle(nullptr);
  }

  solo::KBuffers<T> buffers(const T* actions, uint32_t flags) const {
    solo::KBuffers<T> b;
    b.state = state; b.snapshot = snapshot; b.targets = targets; b.actions = actions;
    b.params = params; b.traj = nullptr; b.obs_inline = b.reward_inline = nullptr; b.done = done; b.term_count = term_count;
    b.obs_rec = b.reward_rec = b.view_obs = b.view_reward = nullptr; b.view_done = nullptr; b.obs_rec_stride = b.reward_rec_stride = 0; b.obs_from = 0;
    b.stats = stats; b.terrain = terrain; b.order = use_order ? order : nullptr; b.cost = cost; b.num_envs = n; b.flags = flags; b.env_base = 0; b.count = n; b.steps = 1;
    b.action_stride = b.done_stride = 0;
    b.queue = nullptr; b.q_rings = 1; b.q_chunk = 0;
    b.fault = fault_dev;
    b.warm = cfg.solver_warm_start > 0 ? warm : nullptr;
#ifdef SOLO_STAMPS
    b.stamps = stamps;
#endif
    return b;
  }

  int launch(const T* actions, uint32_t flags, hipStream_t s) {
    // one 64-lane workgroup (= one wavefront) per robot, one env step
    return launch_chain(Plan{1, 1, 1, 0}, actions, 0, 1, flags, nullptr, nullptr, nullptr, s, 0, n);
  }

  // (SOLO_ERR_INCOMPLETE, sticky: a wave of an earlier migrating launch gave up waiting - a plain read of a pinned host word)
  int check_fault() override {
    if (fault_host != nullptr && *(volatile int32_t*)fault_host != 0) {
      err = "a wave of an earlier launch with robot migration gave up waiting for its robot: some robots were not stepped through that launch (internal error)";
      return SOLO_ERR_INCOMPLETE;
    }
    return SOLO_OK;
  }

  int settle(hipStream_t s) override {
    if (int rc = check_fault()) return rc;
    HIP_TRY(hipSetDevice(device));
    const int total = n * SOLO_STATE_STRIDE;
    hipLaunchKernelGGL(solo::solo_init_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, s, state, targets,
                       (T)cfg.start_pos[0], (T)cfg.start_pos[1], (T)cfg.start_pos[2], (T)cfg.start_quat[0],
                       (T)cfg.start_quat[1], (T)cfg.start_quat[2], (T)cfg.start_quat[3], n);
    HIP_TRY(hipGetLastError());
    // the snapshot doubles as the divergence fallback during the settle loop itself
    HIP_TRY(hipMemcpyAsync(snapshot, state, (size_t)total * sizeof(T), hipMemcpyDeviceToDevice, s));
    // (the settle loop starts from an empty warm-start cache too: a second settle - after a terrain or parameter change -
    // must not depend on what was simulated before)
    HIP_TRY(hipMemsetAsync(warm, 0, (size_t)n * 64 * sizeof(T), s));
    // the settle loop repeats one action: action stride 0 inside the fused launches
    if (int rc = launch_chain(make_plan(cfg.settle_steps, SOLO_STEP_PHYSICS), settle_actions, 0, cfg.settle_steps, SOLO_STEP_PHYSICS, nullptr, nullptr, nullptr, s, 0, n)) return rc;
    HIP_TRY(hipMemcpyAsync(snapshot, state, (size_t)total * sizeof(T), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemsetAsync(term_count, 0, (size_t)n * SOLO_MAX_TERMS * sizeof(int32_t), s));
    HIP_TRY(hipMemsetAsync(warm, 0, (size_t)n * 64 * sizeof(T), s));  // (the snapshot starts from an empty warm-start cache)
    HIP_TRY(hipMemsetAsync(stats, 0, kStatsBytes, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SOLO_OK;
  }

  int set_program(const SoloProgram* p) override {
    HIP_TRY(hipSetDevice(device));
    solo::KParams<T> tmp = hparams;
    if (int rc = solo::pack_program<T>(*p, &tmp, &err)) return rc;
    hparams = tmp;
    obs_dim = p->num_obs;
    have_program = true;
    // uploads are rare (registration time): a blocking copy keeps later launches ordered
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(dparams, &hparams, sizeof(hparams), hipMemcpyHostToDevice));
    return SOLO_OK;
  }

  int reset(const uint8_t* mask, hipStream_t s) override {
    if (int rc = check_fault()) return rc;
    HIP_TRY(hipSetDevice(device));
    const int total = n * SOLO_STATE_STRIDE;
    hipLaunchKernelGGL(solo::solo_reset_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, s, dparams, state, snapshot,
                       targets, term_count, warm, mask, n);
    HIP_TRY(hipGetLastError());
    return SOLO_OK;
  }

  int set_targets(const void* a, hipStream_t s) override {
    if (!a) { err = "actions must not be NULL"; return SOLO_ERR_INVALID_ARG; }
    HIP_TRY(hipSetDevice(device));
    const int total = n * SOLO_NUM_JOINTS;
    hipLaunchKernelGGL(solo::solo_set_targets_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, s, (const T*)a,
                       targets, (T)cfg.action_scale, total);
    HIP_TRY(hipGetLastError());
    return SOLO_OK;
  }

  int check_flags(uint32_t flags) {
    if ((flags & SOLO_STEP_ALL) == 0 || (flags & ~(SOLO_STEP_ALL | SOLO_STEP_AUTO_RESET))) { err = "bad step flags"; return SOLO_ERR_INVALID_ARG; }
    if ((flags & (SOLO_STEP_OBS | SOLO_STEP_REWARD | SOLO_STEP_DONE)) && !have_program) {
      err = "no observation/reward/termination program registered";
      return SOLO_ERR_NO_PROGRAM;
    }
    // mirrors the ValueErrors of obs.py:138-139, rewards.py:115-116, termination.py:43-44
    if ((flags & SOLO_STEP_OBS) && hparams.c.num_obs == 0) { err = "Need to register at least one observation instance"; return SOLO_ERR_NO_PROGRAM; }
    if ((flags & SOLO_STEP_REWARD) && hparams.c.num_reward_ops == 0) { err = "Need to register at least one reward instance"; return SOLO_ERR_NO_PROGRAM; }
    if ((flags & SOLO_STEP_DONE) && hparams.c.num_terms == 0) { err = "Need to register at least one termination instance"; return SOLO_ERR_NO_PROGRAM; }
    return SOLO_OK;
  }

  int step(const void* a, uint32_t flags, hipStream_t s) override {
    if (int rc = check_fault()) return rc;
    if (int rc = check_flags(flags)) return rc;
    HIP_TRY(hipSetDevice(device));
    return launch((const T*)a, flags, s);
  }

  // ---- THE LAUNCH POLICY (round 5: it was bench.py's).  A rollout of k steps runs as `launches` fused launches of S steps
  //      per slice, `slices` independent launch chains, robots migrating every `migrate` steps of a launch (0: never).
  //      Configured values are taken as they are; -1 = the engine chooses, from what was measured on the benchmark workload
  //      (profiles/round5_launch_policy_f32_8192.log, profiles/round5_baseline_configs_f64.log):
  //        * S = min(k, 250): the state record never leaves LDS inside a launch, and 250 steps average out the robots'
  //          unequal solver costs (1.8e8 env-steps/s against 1.2e8 at 20 steps per launch, f64);
  //        * two slices when the rollout takes several launches (one slice's launch boundary and tail overlap the other's
  //          work: +1 ... 2 %), one when it is a single launch (halves of a single launch only shorten each other's tails);
  //        * no migration while every robot of a launch has a wave slot of its own - 4096 robots: four waves on each of the
  //          1024 SIMDs in BOTH precisions since round 5 (f64 round 4: three - 3072 slots - and migration was worth +16 %):
  //          with all robots resident a hand-over only costs; beyond that, in f64, two chunks per launch (several launches:
  //          chunks of 25 steps on ONE chain) let the waves that finish early take over the robots that started late
  //          (f32: never - measured slower).
  struct Plan { int S, launches, slices, migrate; };
  static constexpr int kWavesPerSimd = solo::kWavesPerSimd<T>;
  int resident_robots() const {
    hipDeviceProp_t p;
    const int cus = (hipGetDeviceProperties(&p, device) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    return cus * 4 * kWavesPerSimd;
  }
  int resident_cache = 0;
  Plan make_plan(int k, uint32_t flags) {
    if (resident_cache == 0) resident_cache = resident_robots();
    // steps per fused launch, capped so that the records of one launch stay below 2^32 elements (the step kernel
    // addresses them with 32-bit offsets; at 4096 robots that is 32768 steps)
    const long long cap = ((1ll << 32) - 1) / ((long long)n * SOLO_STATE_STRIDE);
    long long want = cfg.steps_per_launch == -1 ? (k < 250 ? k : 250) : (cfg.steps_per_launch > 1 ? cfg.steps_per_launch : 1);
    if (want > k && k > 0) want = k;
    Plan p;
    p.S = (int)(want < cap ? want : (cap > 1 ? cap : 1));
    if (p.S < 1) p.S = 1;
    p.launches = (k + p.S - 1) / p.S;
    const bool physics_only = flags == SOLO_STEP_PHYSICS;   // (stepSimulation-only launches - the settle loop - never migrate: their robots are in step)
    int streams = cfg.rollout_streams == -1 ? (p.launches > 1 ? 2 : 1) : cfg.rollout_streams;
    streams = streams < 1 ? 1 : (streams > kMaxStreams ? kMaxStreams : streams);
    p.migrate = 0;
#ifndef SOLO_STAMPS   // (never in the diagnostic stamps builds, whose per-wave stamps assume one robot per wave)
    if (cfg.migrate_steps > 0) p.migrate = cfg.migrate_steps;
    else if (cfg.migrate_steps == -1 && !physics_only && (flags & SOLO_STEP_PHYSICS) && p.S >= 8) {
      // (8192 robots, f64, profiles/round5_baseline_configs_f64.log: one launch of 20 steps 1.485e8 in two chunks against
      // 1.474e8 without; 1000 steps 1.995e8 as one chain in chunks of 25 against 1.956e8 on two slices, 1.76e8 on one chain)
      // f32 at 8192 robots: migration costs 2.6 % (K = 20) / 3.8 % (1000 steps) - its robot-steps are short against a
      // hand-over (profiles/round5_launch_policy_f32_8192.log): f64 only
      if (n > resident_cache && sizeof(T) == 8) {
        if (p.launches == 1) p.migrate = (p.S + 1) / 2;
        else if (p.S >= 50) { p.migrate = 25; if (cfg.rollout_streams == -1) streams = 1; }
      }
    }
#endif
    p.slices = (streams > 1 && n >= 2 * streams) ? streams : 1;
    return p;
  }
  // ints of the migration queue per robot (its sweep counter + its ring slots), for EVERY launch of a chain whose launches
  // fuse up to S steps: the chunk count is not monotone in the step count - migration_chunk_steps stretches the chunks of a
  // launch that would need more than 127 (a ring slot has 7 bits for the chunk index), so a ragged last launch of fewer steps
  // can need MORE chunks than the full ones (S = 128, migrate 1: 64 chunks of 2; a 65-step tail: 65 chunks of 1) - but no
  // launch ever has more than min(127, ceil(S / migrate)) (the regions were sized and strided for the S-step
  // launch's count, and such a tail wrote past its slice's region).  The same expression sizes the allocation and strides
  // the slices' regions (launch_chain), which asserts that every launch's queue fits.
  static int queue_slots_per_robot(const Plan& p) {
    const int chunks = solo::migration_chunks(p.S, p.migrate);
    return 1 + (chunks < 127 ? chunks : 127);
  }
  // the record scratch of fused launches and the migration queues are sized for the rollout at hand (a larger one grows them:
  // the one hidden device synchronisation of the stream-ordered calls - include/solo_engine.h "LAZY SCRATCH"; solo_engine_reserve
  // does it ahead of time)
  int ensure_scratch(const Plan& p, uint32_t flags) {
    const bool records = (flags & (SOLO_STEP_OBS | SOLO_STEP_REWARD)) != 0;
    if (records && p.S > traj_steps) {
      HIP_TRY(hipDeviceSynchronize());
      if (traj) { (void)hipFree(traj); traj = nullptr; traj_steps = 0; }
      HIP_TRY(hipMalloc((void**)&traj, (size_t)p.S * (size_t)n * SOLO_STATE_STRIDE * sizeof(T)));
      traj_steps = p.S;
    }
    if (p.migrate > 0) {
      const size_t need = (size_t)kMaxStreams * solo::kQueueHeader + (size_t)n * (size_t)queue_slots_per_robot(p);
      if (need > queue_ints) {
        HIP_TRY(hipDeviceSynchronize());
        if (queue) { (void)hipFree(queue); queue = nullptr; queue_ints = 0; }
        HIP_TRY(hipMalloc((void**)&queue, need * sizeof(int32_t)));
        queue_ints = need;
      }
    }
    return SOLO_OK;
  }

  // one chain of launches covering steps [0, k) for robots [lo, lo+count): per launch the step
  // kernel (one wave per robot, S fused steps; its output epilogue evaluates the S step records the robot left)
  // final_chunk: this chain ends the caller's rollout - a RECORDING rollout then has its last launch's epilogue
  // also leave the last step's observation / reward / done in the engine's view (what three
  // device-to-device copies after the chain used to do: ~15 us of a 0.4 ms 20-step rollout)
  int launch_chain(const Plan& plan, const T* act, long long act_stride, int k, uint32_t flags, T* obs_out, T* reward_out,
                   uint8_t* done_out, hipStream_t s, int lo, int count, bool final_chunk = true, int slice = 0) {
    const int S = plan.S;
    if (int rc = ensure_scratch(plan, flags)) return rc;
    const bool want_obs = (flags & SOLO_STEP_OBS) != 0, want_reward = (flags & SOLO_STEP_REWARD) != 0;
    for (int i = 0; i < k; i += S) {
      const int steps = (k - i < S) ? (k - i) : S;
      solo::KBuffers<T> b = buffers(act ? act + (size_t)i * act_stride : nullptr, flags);
      b.env_base = lo;
      b.count = count;
      b.steps = steps;
      b.action_stride = act_stride;
      if (done_out) { b.done = done_out + (size_t)i * n; b.done_stride = n; }
      // a single-step f32 launch (closed-loop step(), or a rollout with steps_per_launch = 1) evaluates its outputs
      // lane-parallel over the items of the one step; every other launch leaves records, which the robot's wave
      // evaluates after its last step (the output epilogue of the step kernel): where they go -
      const bool inline_outputs = steps == 1 && (want_obs || want_reward) && solo::kInlineOutputs<T, true>;
      if (inline_outputs) {
        if (want_obs) b.obs_inline = obs_out ? obs_out + (size_t)i * n * obs_dim : obs;
        if (want_reward) b.reward_inline = reward_out ? reward_out + (size_t)i * n : reward;
      } else if (want_obs || want_reward) {
        // (the launch's own region of the record scratch, indexed by robot - first robot of the launch: the slices of
        // a rollout run side by side with DIFFERENT step counts when one of them is already in its ragged last launch -
        // indexed by absolute robot x steps of the launch their regions overlapped: a race that round 4's warm-start
        // test caught, tests/test_gpu_warm_start.py)
        b.traj = traj + (size_t)lo * (size_t)traj_steps * SOLO_STATE_STRIDE;
        // a recording rollout keeps every step ([K][N][.] buffers of the caller) and its last launch also leaves the
        // last step in the engine's view; otherwise only the last step's observation / reward / done stay in the view
        const bool tail_to_view = final_chunk && i + S >= k;
        if (want_obs) {
          if (obs_out) { b.obs_rec = obs_out + (size_t)i * n * obs_dim; b.obs_rec_stride = (long long)n * obs_dim; if (tail_to_view) b.view_obs = obs; }
          else b.view_obs = obs;
        }
        if (want_reward) {
          if (reward_out) { b.reward_rec = reward_out + (size_t)i * n; b.reward_rec_stride = n; if (tail_to_view) b.view_reward = reward; }
          else b.view_reward = reward;
        }
        if (done_out && (flags & SOLO_STEP_DONE) && tail_to_view) b.view_done = done;
      }
      // robot migration: a launch of more than one chunk of steps gets a work queue (its own region per rollout slice:
      // the slices' launches run side by side), initialised on the stream in front of the step kernel; eight rings -
      // one per XCD - when the robots divide evenly, else one
      // (stepSimulation-only launches - the settle loop, client.stepSimulation() - keep the physics-only instantiation:
      // their robots are in step with each other, there is nothing to balance, and in profiles the settle loop stays a
      // kernel of its own instead of inflating the measured one's average)
      if (plan.migrate > 0 && steps > plan.migrate && (flags & SOLO_STEP_PHYSICS) && flags != SOLO_STEP_PHYSICS) {
        const int chunk = solo::migration_chunk_steps(steps, plan.migrate);
        b.q_chunk = chunk;
        b.q_rings = solo::migration_rings(count);
        b.queue = queue + (size_t)slice * solo::kQueueHeader + (size_t)lo * (size_t)queue_slots_per_robot(plan);
        const size_t ints = solo::migration_queue_ints(count, steps, chunk);
        if (ints > (size_t)solo::kQueueHeader + (size_t)count * (size_t)queue_slots_per_robot(plan) ||
            (size_t)(b.queue - queue) + ints > queue_ints) {   // (never: queue_slots_per_robot bounds every launch of <= S steps)
          err = "internal error: the migration queue of a launch does not fit its slice's region";
          return SOLO_ERR_INVALID_ARG;
        }
        hipLaunchKernelGGL(solo::solo_queue_init_kernel, dim3((unsigned)((ints + 255) / 256)), dim3(256), 0, s, b.queue, ints, lo, count,
                           b.q_rings, steps, chunk, (const int32_t*)(use_order ? order : nullptr));
        HIP_TRY(hipGetLastError());
      }
      // stepSimulation-only calls (settle loop, client.stepSimulation()) run the physics-only
      // instantiation: no termination code, and a separate name in profiles
      // (pybullet's residual threshold, an opt-in, is a kernel instantiation of its own: the default kernels carry none of it)
      const bool resid = cfg.solver_residual_threshold > 0;
      if (b.queue != nullptr) {  // (robot migration is a kernel instantiation of its own too - always the full kernel: kFull only selects code)
        if (resid) hipLaunchKernelGGL((solo::solo_step_kernel<T, true, true, true>), dim3(count), dim3(64), 0, s, dparams, b);
        else hipLaunchKernelGGL((solo::solo_step_kernel<T, true, false, true>), dim3(count), dim3(64), 0, s, dparams, b);
      } else if (flags == SOLO_STEP_PHYSICS) {
        if (resid) hipLaunchKernelGGL((solo::solo_step_kernel<T, false, true>), dim3(count), dim3(64), 0, s, dparams, b);
        else hipLaunchKernelGGL((solo::solo_step_kernel<T, false, false>), dim3(count), dim3(64), 0, s, dparams, b);
      } else {
        if (resid) hipLaunchKernelGGL((solo::solo_step_kernel<T, true, true>), dim3(count), dim3(64), 0, s, dparams, b);
        else hipLaunchKernelGGL((solo::solo_step_kernel<T, true, false>), dim3(count), dim3(64), 0, s, dparams, b);
      }
      HIP_TRY(hipGetLastError());
    }
    return SOLO_OK;
  }

  int rollout(const void* a, int k, uint32_t flags, void* obs_out, void* reward_out, void* done_out, hipStream_t s) override {
    if (int rc = check_fault()) return rc;
    if (int rc = check_flags(flags)) return rc;
    if (k == 0) return SOLO_OK;  // an empty rollout is a no-op
    if (!a || k < 0) { err = "rollout needs actions [K][N][12]"; return SOLO_ERR_INVALID_ARG; }
    const Plan plan = make_plan(k, flags);
    if (int rc = rollout_impl(plan, (const T*)a, k, flags, obs_out, reward_out, done_out, s, nullptr, nullptr)) return rc;
    // the engine's view always ends up with the LAST step's outputs, also when every step was recorded: the last
    // launch's output kernel writes them (launch_chain); only rollouts whose last launch is a single-step launch with
    // in-place outputs (steps_per_launch = 1 in f32, or a one-step remainder) copy
    const int S = plan.S;
    const int last_steps = (k % S == 0) ? S : k % S;
    // ... and recording rollouts that asked for neither observations nor rewards: their launches leave no records (no
    // epilogue runs), the step kernel writes the done flags straight into the caller's [K][N] buffer
    const bool no_epilogue = (flags & (SOLO_STEP_OBS | SOLO_STEP_REWARD)) == 0;
    if (!(no_epilogue || (last_steps == 1 && solo::kInlineOutputs<T, true>))) return SOLO_OK;
    if (obs_out && (flags & SOLO_STEP_OBS))
      HIP_TRY(hipMemcpyAsync(obs, (const T*)obs_out + (size_t)(k - 1) * n * obs_dim, (size_t)n * obs_dim * sizeof(T), hipMemcpyDeviceToDevice, s));
    if (reward_out && (flags & SOLO_STEP_REWARD))
      HIP_TRY(hipMemcpyAsync(reward, (const T*)reward_out + (size_t)(k - 1) * n, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, s));
    if (done_out && (flags & SOLO_STEP_DONE))
      HIP_TRY(hipMemcpyAsync(done, (const uint8_t*)done_out + (size_t)(k - 1) * n, (size_t)n, hipMemcpyDeviceToDevice, s));
    return SOLO_OK;
  }

  // t0 / t1 (optional, [groups] each): timing events recorded around every slice's launch chain, on
  // the stream that chain is launched on
  int rollout_impl(const Plan& plan, const T* act, int k, uint32_t flags, void* obs_out, void* reward_out, void* done_out,
                   hipStream_t s, hipEvent_t* t0, hipEvent_t* t1, int* groups_out = nullptr) {
    HIP_TRY(hipSetDevice(device));
    const long long stride = act ? (long long)n * SOLO_NUM_JOINTS : 0;
    T* oo = (flags & SOLO_STEP_OBS) ? (T*)obs_out : nullptr;
    T* ro = (flags & SOLO_STEP_REWARD) ? (T*)reward_out : nullptr;
    uint8_t* dn = (flags & SOLO_STEP_DONE) ? (uint8_t*)done_out : nullptr;
    const int groups = plan.slices;
    if (groups_out) *groups_out = groups;
    if (int rc = ensure_scratch(plan, flags)) return rc;   // (before any timing event is recorded)
    if (groups == 1) {
      if (t0) HIP_TRY(hipEventRecord(t0[0], s));
      if (int rc = launch_chain(plan, act, stride, k, flags, oo, ro, dn, s, 0, n)) return rc;
      if (t1) HIP_TRY(hipEventRecord(t1[0], s));
      return SOLO_OK;
    }
    // Robots are independent, so the batch can be cut into `groups` slices that advance through
    // the K steps as independent launch chains on their own HIP streams: one slice's kernel
    // boundary / tail overlaps the other slices' work.  Fork from and join into the caller's stream.
    if (int rc = ensure_streams(groups)) return rc;
    HIP_TRY(hipEventRecord(ev_fork, s));
    for (int g = 0; g < groups; ++g) {
      HIP_TRY(hipStreamWaitEvent(sub[g], ev_fork, 0));
      if (t0) HIP_TRY(hipEventRecord(t0[g], sub[g]));
    }
    const int S = plan.S;
    for (int i = 0; i < k; i += S)
      for (int g = 0; g < groups; ++g) {
        const int lo = (int)((long long)n * g / groups), hi = (int)((long long)n * (g + 1) / groups);
        const int kk = (k - i < S) ? (k - i) : S;
        if (int rc = launch_chain(plan, act ? act + (size_t)i * stride : nullptr, stride, kk, flags,
                                  oo ? oo + (size_t)i * n * obs_dim : nullptr, ro ? ro + (size_t)i * n : nullptr,
                                  dn ? dn + (size_t)i * n : nullptr, sub[g], lo, hi - lo, i + S >= k, g))
          return rc;
      }
    for (int g = 0; g < groups; ++g) {
      if (t1) HIP_TRY(hipEventRecord(t1[g], sub[g]));
      HIP_TRY(hipEventRecord(ev_join[g], sub[g]));
      HIP_TRY(hipStreamWaitEvent(s, ev_join[g], 0));
    }
    return SOLO_OK;
  }

  // the most slices a rollout of this engine can be cut into (what a launch order has to respect)
  int max_slices() const {
    const int streams = cfg.rollout_streams == -1 ? 2 : (cfg.rollout_streams < 1 ? 1 : (cfg.rollout_streams > kMaxStreams ? kMaxStreams : cfg.rollout_streams));
    return (streams > 1 && n >= 2 * streams) ? streams : 1;
  }

  static constexpr int kMaxStreams = 8;
  hipStream_t sub[kMaxStreams] = {};
  hipEvent_t ev_fork = nullptr, ev_join[kMaxStreams] = {};
  int ensure_streams(int groups) {
    if (!ev_fork) HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    for (int g = 0; g < groups; ++g) {
      if (!sub[g]) HIP_TRY(hipStreamCreateWithFlags(&sub[g], hipStreamNonBlocking));
      if (!ev_join[g]) HIP_TRY(hipEventCreateWithFlags(&ev_join[g], hipEventDisableTiming));
    }
    return SOLO_OK;
  }

  int time_step(const void* a, uint32_t flags, int reps, hipStream_t s, double* ms) override {
    if (int rc = check_flags(flags)) return rc;
    if (reps <= 0 || !ms) { err = "reps must be positive"; return SOLO_ERR_INVALID_ARG; }
    HIP_TRY(hipSetDevice(device));
    struct Events {  // destroyed on every path out of this function
      hipEvent_t e[2 * kMaxStreams] = {};
      ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
    } ev;
    for (hipEvent_t& x : ev.e) HIP_TRY(hipEventCreate(&x));
    hipEvent_t* e0 = ev.e;
    hipEvent_t* e1 = ev.e + kMaxStreams;
    int groups = 1;
    // (reps launches of the CONFIGURED steps per launch - one step when that is left to the engine: solo_engine_time_rollout
    // times a rollout of a given length with the engine's own geometry)
    const int S = cfg.steps_per_launch > 1 ? cfg.steps_per_launch : 1;
    Plan plan = make_plan(reps * S, flags);
    if (cfg.steps_per_launch == -1) {   // (reps launches of ONE step, whatever the engine would choose for a rollout of reps steps - on ONE chain, whatever reps is)
      plan.S = 1; plan.launches = reps; plan.migrate = 0;
      if (cfg.rollout_streams == -1) plan.slices = 1;
    }
    const int rc_chain = rollout_impl(plan, (const T*)a, reps * S, flags, nullptr, nullptr, nullptr, s, e0, e1, &groups);
    if (rc_chain) return rc_chain;
    HIP_TRY(hipStreamSynchronize(s));
    // every slice's chain is timed on its own stream; the AVERAGE launch duration over all slices
    // and launches is reported (the statistic rocprofv3 --stats gives for the kernel)
    double total = 0;
    for (int g = 0; g < groups; ++g) {
      float t = 0;
      HIP_TRY(hipEventElapsedTime(&t, e0[g], e1[g]));
      total += t;
    }
    *ms = total / groups / plan.launches;
    return SOLO_OK;
  }

  int time_rollout(const void* a, int k, uint32_t flags, void* obs_out, void* reward_out, void* done_out, hipStream_t s, double* ms) override {
    if (int rc = check_flags(flags)) return rc;
    if (k <= 0 || !ms) { err = "num_steps must be positive"; return SOLO_ERR_INVALID_ARG; }
    HIP_TRY(hipSetDevice(device));
    struct Events {
      hipEvent_t e[2 * kMaxStreams] = {};
      ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
    } ev;
    for (hipEvent_t& x : ev.e) HIP_TRY(hipEventCreate(&x));
    int groups = 1;
    const Plan plan = make_plan(k, flags);
    if (int rc = rollout_impl(plan, (const T*)a, k, flags, obs_out, reward_out, done_out, s, ev.e, ev.e + kMaxStreams, &groups)) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    double total = 0;
    for (int g = 0; g < groups; ++g) {
      float t = 0;
      HIP_TRY(hipEventElapsedTime(&t, ev.e[g], ev.e[kMaxStreams + g]));
      total += t;
    }
    *ms = total / groups / plan.launches;
    return SOLO_OK;
  }

  int reserve(int k, uint32_t flags) override {
    if (k <= 0) { err = "num_steps must be positive"; return SOLO_ERR_INVALID_ARG; }
    if (int rc = check_flags(flags)) return rc;
    HIP_TRY(hipSetDevice(device));
    // every rollout of up to k steps under this configuration: steps per launch grow with the rollout up to the cap, and the
    // migration policy depends on the number of launches - the geometries of 1 .. k steps reduce to a handful
    int last_S = -1, last_m = -1;
    for (int kk = 1; kk <= k; kk = (kk < 512 ? kk + 1 : (kk * 2 < k ? kk * 2 : (kk == k ? k + 1 : k)))) {
      const Plan p = make_plan(kk, flags);
      if (p.S == last_S && p.migrate == last_m) continue;
      last_S = p.S; last_m = p.migrate;
      if (int rc = ensure_scratch(p, flags)) return rc;
      if (p.slices > 1) if (int rc = ensure_streams(p.slices)) return rc;
    }
    return SOLO_OK;
  }

  int plan(int k, SoloLaunchPlan* out) override {
    if (k <= 0 || !out) { err = "num_steps must be positive"; return SOLO_ERR_INVALID_ARG; }
    HIP_TRY(hipSetDevice(device));
    const Plan p = make_plan(k, SOLO_STEP_ALL);
    out->steps_per_launch = p.S; out->launches = p.launches; out->slices = p.slices;
    out->migrate_steps = (p.migrate > 0 && p.S > p.migrate) ? solo::migration_chunk_steps(p.S, p.migrate) : 0;
    out->waves_per_simd = kWavesPerSimd; out->resident_robots = resident_cache;
    return SOLO_OK;
  }

  int set_terrain(const SoloTerrain* t, hipStream_t s) override {
    // (validated before anything is touched: a rejected call leaves the previous ground in force)
    if (t && (t->nx < 2 || t->ny < 2 || !(t->cell > 0) || !t->heights || (long long)t->nx * t->ny > (1ll << 26))) {
      err = "terrain needs nx, ny >= 2, cell > 0 and a heights array";
      return SOLO_ERR_INVALID_ARG;
    }
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    if (terrain) { (void)hipFree(terrain); terrain = nullptr; }
    hparams.c.terr_nx = hparams.c.terr_ny = 0;
    if (t) {
      const size_t cnt = (size_t)t->nx * t->ny;
      std::vector<T> h(cnt);
      for (size_t i = 0; i < cnt; ++i) h[i] = (T)t->heights[i];
      HIP_TRY(hipMalloc((void**)&terrain, cnt * sizeof(T)));
      HIP_TRY(hipMemcpy(terrain, h.data(), cnt * sizeof(T), hipMemcpyHostToDevice));
      hparams.c.terr_nx = t->nx; hparams.c.terr_ny = t->ny;
      hparams.c.terr_inv_cell = (T)(1.0 / t->cell);
      hparams.c.terr_ox = (T)t->origin[0]; hparams.c.terr_oy = (T)t->origin[1];
    }
    HIP_TRY(hipMemcpy(dparams, &hparams, sizeof(hparams), hipMemcpyHostToDevice));
    return settle(s);
  }

  int set_order(const int32_t* o, hipStream_t s) override {
    HIP_TRY(hipSetDevice(device));
    if (o == nullptr) { use_order = false; return SOLO_OK; }
    // The step kernel uses order[slot] as the robot index of every buffer it addresses, and the output epilogues
    // of a rollout slice read the records of that slice's robots: the table is validated here, once per upload
    // (a rare call; blocking) - a permutation of [0, N) that maps every rollout slice onto itself.
    std::vector<int32_t> h((size_t)n);
    HIP_TRY(hipMemcpyAsync(h.data(), o, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<uint8_t> seen((size_t)n, 0);
    const int groups = max_slices();
    for (int g = 0; g < groups; ++g) {
      const int lo = (int)((long long)n * g / groups), hi = (int)((long long)n * (g + 1) / groups);
      for (int i = lo; i < hi; ++i) {
        const int32_t e = h[(size_t)i];
        if (e < lo || e >= hi || seen[(size_t)e]) {
          err = "launch order must be a permutation of [0, num_envs) that keeps every rollout slice's robots in that slice";
          return SOLO_ERR_INVALID_ARG;
        }
        seen[(size_t)e] = 1;
      }
    }
    HIP_TRY(hipMemcpyAsync(order, h.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // (h goes out of scope)
    use_order = true;
    return SOLO_OK;
  }

  int view(SoloStateView* v) override {
    v->num_envs = n;
    v->dtype = sizeof(T) == 4 ? SOLO_F32 : SOLO_F64;
    v->state_stride = SOLO_STATE_STRIDE;
    v->obs_dim = obs_dim;
    v->state = state; v->snapshot = snapshot; v->targets = targets; v->obs = obs; v->reward = reward;
    v->done = done; v->term_count = term_count; v->params = params; v->stats = stats; v->cost = cost; v->warm = warm;
    return SOLO_OK;
  }

  int set_params(int which, const void* p, hipStream_t s) override {
    if (which < 0 || which > 1 || !p) { err = "which must be 0 (friction) or 1 (base mass scale)"; return SOLO_ERR_INVALID_ARG; }
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy2DAsync(params + which, 4 * sizeof(T), p, sizeof(T), sizeof(T), n, hipMemcpyDeviceToDevice, s));
    return SOLO_OK;
  }

  const char* kernel_name() override {
    const bool resid = cfg.solver_residual_threshold > 0;
    // (the instantiation of a launch whose robots do not migrate - every launch of up to 4096 robots under the engine's own
    // policy; a migrating launch's last template argument is `true`)
    if (sizeof(T) == 4) return resid ? "solo_step_kernel<float, true, true, false>" : "solo_step_kernel<float, true, false, false>";
    return resid ? "solo_step_kernel<double, true, true, false>" : "solo_step_kernel<double, true, false, false>";
  }
};

// the f64 engine's factory: the one entry point of its translation unit (null = out of host memory; *rc = init's status)
EngineBase* make_engine_f64(const SoloConfig& cfg, const SoloModel& model, int32_t num_envs, int32_t device_id, int* rc);
#ifndef SOLO_TU_F32
EngineBase* make_engine_f64(const SoloConfig& cfg, const SoloModel& model, int32_t num_envs, int32_t device_id, int* rc) {
  auto* e = new (std::nothrow) Engine<double>();
  *rc = e ? e->init(cfg, model, num_envs, device_id) : (int)SOLO_ERR_HIP;
  return e;
}
#endif

}  // namespace solo_engine_detail

#ifndef SOLO_TU_F64  // ---- the C ABI (and the f32 engine) ----
using namespace solo_engine_detail;

namespace {

thread_local std::string g_create_error;

int check_config(const SoloConfig* c, std::string* err) {
  auto fail = [&](const char* s) { *err = s; return (int)SOLO_ERR_INVALID_ARG; };
  if (c->abi_version != SOLO_ABI_VERSION) return fail("abi_version mismatch");
  if (c->dtype != SOLO_F32 && c->dtype != SOLO_F64) return fail("dtype must be SOLO_F32 or SOLO_F64");
  if (!(c->dt > 0)) return fail("dt must be positive");
  if (c->solver_iterations < 1 || c->solver_iterations > 10000) return fail("solver_iterations out of range");
  if (c->solver_ulp_tolerance < 0 || c->solver_ulp_tolerance > (1 << 20)) return fail("solver_ulp_tolerance out of range");
  if (!(c->solver_residual_threshold >= 0)) return fail("solver_residual_threshold must be >= 0");
  if (c->settle_steps < 0 || c->settle_steps > 100000) return fail("settle_steps out of range");
  if (c->steps_per_launch < -1 || c->steps_per_launch > 100000) return fail("steps_per_launch out of range (-1 = the engine chooses)");
  if (c->rollout_streams < -1 || c->rollout_streams > 64) return fail("rollout_streams out of range (-1 = the engine chooses)");
  if (!(c->solver_warm_start >= 0 && c->solver_warm_start <= 1)) return fail("solver_warm_start must be in [0, 1]");
  if (c->solver_warm_start > 0 && !(c->solver_residual_threshold > 0)) return fail("solver_warm_start needs solver_residual_threshold > 0");
  if (c->migrate_steps < -1 || c->migrate_steps > 100000) return fail("migrate_steps out of range (-1 = the engine chooses)");
  if (c->reserved0 != 0) return fail("reserved0 must be 0");
  // (restitution: accepted, and without effect - see include/solo_engine.h: Bullet combines it with the ground's, which is 0)
  if (!(c->restitution >= 0 && c->restitution <= 1)) return fail("restitution must be in [0, 1] (gym_solo configs.py:23)");
  if (!(c->action_scale > 0)) return fail("action_scale must be positive");
  if (c->lateral_friction < 0 || c->contact_margin < 0 || c->contact_erp < 0) return fail("negative contact parameter");
  if (!(c->base_lateral_friction >= 0)) return fail("base_lateral_friction must be >= 0");
  return SOLO_OK;
}

}  // namespace

struct SoloEngine { EngineBase* impl; };

extern "C" {

#ifdef SOLO_STAMPS
// DIAGNOSTIC build only: copies the [N][32] s_memtime stamps of the last launch to the host.
int solo_engine_debug_stamps(SoloEngine* eng, unsigned long long* host, int is_f32) {
  if (!eng || !eng->impl) return SOLO_ERR_INVALID_ARG;
  (void)hipDeviceSynchronize();
  if (is_f32) { auto* e = static_cast<Engine<float>*>(eng->impl); return hipMemcpy(host, e->stamps, (size_t)e->n * 256, hipMemcpyDeviceToHost) == hipSuccess ? 0 : SOLO_ERR_HIP; }
  auto* e = static_cast<Engine<double>*>(eng->impl);
  return hipMemcpy(host, e->stamps, (size_t)e->n * 256, hipMemcpyDeviceToHost) == hipSuccess ? 0 : SOLO_ERR_HIP;
}
#endif

int solo_abi_version(void) { return SOLO_ABI_VERSION; }
const char* solo_last_create_error(void) { return g_create_error.c_str(); }

int solo_engine_create(const SoloConfig* cfg, const SoloModel* model, int32_t num_envs, int32_t device_id,
                       SoloEngine** out) {
  if (!cfg || !model || !out) { g_create_error = "NULL argument"; return SOLO_ERR_INVALID_ARG; }
  *out = nullptr;
  if (num_envs < 1 || num_envs > (1 << 24)) { g_create_error = "num_envs out of range"; return SOLO_ERR_INVALID_ARG; }
  if (int rc = check_config(cfg, &g_create_error)) return rc;
  if (int rc = solo::validate_model(*model, &g_create_error)) return rc;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
    g_create_error = "no HIP device visible: the Solo8 engine has no CPU fallback";
    return SOLO_ERR_NO_DEVICE;
  }
  if (device_id < 0 || device_id >= count) { g_create_error = "device_id out of range"; return SOLO_ERR_INVALID_ARG; }
  EngineBase* impl = nullptr;
  int rc;
  if (cfg->dtype == SOLO_F32) {
    auto* e = new (std::nothrow) Engine<float>();
    impl = e;
    rc = e ? e->init(*cfg, *model, num_envs, device_id) : SOLO_ERR_HIP;
  } else {
    impl = make_engine_f64(*cfg, *model, num_envs, device_id, &rc);
  }
  if (rc != SOLO_OK) {
    g_create_error = impl ? impl->err : "out of host memory";
    delete impl;
    return rc;
  }
  *out = new SoloEngine{impl};
  return SOLO_OK;
}

int solo_engine_destroy(SoloEngine* eng) {
  if (!eng) return SOLO_ERR_INVALID_ARG;
  delete eng->impl;
  delete eng;
  return SOLO_OK;
}

// (SOLO_ERR_INCOMPLETE is sticky: EVERY entry point that takes the engine checks the fault word first - ADVICE r5)
static int eng_fault(SoloEngine* eng) { return eng && eng->impl ? eng->impl->check_fault() : (int)SOLO_ERR_INVALID_ARG; }
#define ENG_CALL(expr) (eng_fault(eng) != SOLO_OK ? eng_fault(eng) : (eng->impl->expr))

int solo_engine_set_program(SoloEngine* eng, const SoloProgram* prog) {
  if (!prog) return SOLO_ERR_INVALID_ARG;
  return ENG_CALL(set_program(prog));
}
int solo_engine_reset(SoloEngine* eng, const uint8_t* mask_dev, void* stream) { return ENG_CALL(reset(mask_dev, (hipStream_t)stream)); }
int solo_engine_settle(SoloEngine* eng, void* stream) { return ENG_CALL(settle((hipStream_t)stream)); }
int solo_engine_set_targets(SoloEngine* eng, const void* a, void* stream) { return ENG_CALL(set_targets(a, (hipStream_t)stream)); }
int solo_engine_step(SoloEngine* eng, const void* a, uint32_t flags, void* stream) { return ENG_CALL(step(a, flags, (hipStream_t)stream)); }
int solo_engine_rollout(SoloEngine* eng, const void* a, int32_t k, uint32_t flags, void* stream) {
  return ENG_CALL(rollout(a, k, flags, nullptr, nullptr, nullptr, (hipStream_t)stream));
}
int solo_engine_rollout_record(SoloEngine* eng, const void* a, int32_t k, uint32_t flags, void* obs_out, void* reward_out,
                               void* done_out, void* stream) {
  return ENG_CALL(rollout(a, k, flags, obs_out, reward_out, done_out, (hipStream_t)stream));
}
int solo_engine_get_view(SoloEngine* eng, SoloStateView* out) {
  if (!out) return SOLO_ERR_INVALID_ARG;
  return ENG_CALL(view(out));
}
int solo_engine_set_params(SoloEngine* eng, int32_t which, const void* p, void* stream) { return ENG_CALL(set_params(which, p, (hipStream_t)stream)); }
int solo_engine_set_terrain(SoloEngine* eng, const SoloTerrain* t, void* stream) { return ENG_CALL(set_terrain(t, (hipStream_t)stream)); }
int solo_engine_set_order(SoloEngine* eng, const int32_t* o, void* stream) { return ENG_CALL(set_order(o, (hipStream_t)stream)); }
const char* solo_engine_kernel_name(SoloEngine* eng) { return eng && eng->impl ? eng->impl->kernel_name() : ""; }
int solo_engine_time_step(SoloEngine* eng, const void* a, uint32_t flags, int32_t reps, void* stream, double* ms) {
  return ENG_CALL(time_step(a, flags, reps, (hipStream_t)stream, ms));
}
int solo_engine_time_rollout(SoloEngine* eng, const void* a, int32_t k, uint32_t flags, void* obs_out, void* reward_out, void* done_out,
                             void* stream, double* ms) {
  return ENG_CALL(time_rollout(a, k, flags, obs_out, reward_out, done_out, (hipStream_t)stream, ms));
}
int solo_engine_plan(SoloEngine* eng, int32_t k, SoloLaunchPlan* out) { return ENG_CALL(plan(k, out)); }
int solo_engine_reserve(SoloEngine* eng, int32_t k, uint32_t flags) { return ENG_CALL(reserve(k, flags)); }
const char* solo_engine_last_error(SoloEngine* eng) { return eng && eng->impl ? eng->impl->err.c_str() : "invalid engine handle"; }

}  // extern "C"
#endif  // !SOLO_TU_F64
