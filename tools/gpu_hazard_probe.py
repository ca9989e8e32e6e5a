"""DIAGNOSTIC (round 3): what does the s_set_gpr_idx_on adjacency failure depend on, ON THE STEP KERNEL?
Runs the identical-robots check (every robot the same state and actions: all must end in the same bits) on the
diagnostic builds of `make -C gym_solo_amd/csrc hazard-probes` (-DSOLO_PGS_HAZARD_PROBE=n, solo_pgs_gfx950.h):
  0 the product's order            1 round 2's order (indexed v_fma directly behind s_set_gpr_idx_on)
  2 round 2's + s_nop behind s_set_gpr_idx_on      3 ... + s_nop behind s_set_gpr_idx_off      4 ... + s_nop in front of it
  5 the cursor shift behind s_set_gpr_idx_on only  6 s_set_gpr_idx_on / indexed v_mov_b32 / off, v_fma on the moved value
  7 as 6 + s_nop                   8 / 9 round 2's order, accumulator + broadcast pinned to v8 / s66 and v7 / s64
  libsolo_hip_probe_<v>_<s>.so     round 2's order with any pinned pair
  round 6, accumulator pinned to v7 (the failing register): 11 v_nop in the shadow   12 the column as source 1 (gpr_idx(SRC1))
  13 the VOP2 encoding (v_fmac_f32)   14 s_nop 7 in FRONT of the switch
in the residual-threshold kernels (which failed in round 3) and in the default ones.  Every build also counts waves
that leave the loop with the index mode still ON (statistics slot 7) and switches it off there.  2048 robots: no
variant listed here has ever faulted (one that is NOT built any more - three VGPR sources on the indexed v_fma - did).
Results: profiles/round3_hazard_probe.log, DESIGN.md section 4.  One process per build (the library is chosen at import):
usage: SOLO_HIP_LIB=.../libsolo_hip_probeN.so python tools/gpu_hazard_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from gym_solo_amd import abi
from gym_solo_amd.envs.solo8v2vanilla import Solo8VanillaConfig, Solo8VanillaEnv
from gym_solo_amd.workloads import register_benchmark_workload

N = int(os.environ.get('N', '2048'))
if N > 2048:
  # round 6: probe 9 (and the VOP2 variant, 13) ended in GPU memory-access FAULTS at 4096 and 8192 robots - garbage columns drive
  # the states to values whose terrain / table addresses are wild; a fault can reset the whole host's GPUs: never again above 2048
  raise SystemExit('tools/gpu_hazard_probe.py: N > 2048 is refused (failing probes fault the GPU at 4096 robots: profiles/round6_hazard_probe.log)')
g = torch.Generator(device='cuda').manual_seed(8)
one = (torch.rand(8, 1, 12, device='cuda', generator=g) * 2 - 1) * (2 * np.pi)
for resid in (1e-7, 0.0):
  for rep in range(2):
    cfg = Solo8VanillaConfig()
    cfg.dtype, cfg.num_envs, cfg.auto_reset, cfg.steps_per_launch, cfg.solver_residual_threshold = 'float32', N, True, 8, resid
    env = Solo8VanillaEnv(config=cfg, copy_outputs=False)
    register_benchmark_workload(env, max_steps=1000)
    env._ensure_program()
    eng = env.engine
    snap = eng.snapshot.cpu().numpy()
    settle_diff = int((snap != snap[0]).any(axis=1).sum())
    eng.rollout(one.expand(8, N, 12).contiguous(), abi.STEP_ALL)
    eng.synchronize()
    st = eng.state.cpu().numpy()
    finite = bool(np.isfinite(st[:, :29]).all())
    diff = int((st != st[0]).any(axis=1).sum())
    stuck = float(eng.stats_shards.cpu().numpy()[:, 7].sum())
    rows, counts = np.unique(st[:, :29].view(np.uint32), axis=0, return_counts=True)
    import zlib
    common = rows[np.argmax(counts)]
    crc = lambda r: zlib.crc32(np.ascontiguousarray(r).tobytes()) & 0xffffffff
    print('%s resid %g rep %d: robots differing from robot 0 after the settle loop %d, after 8 steps %d of %d; waves leaving the loop with the index mode on: %d; finite %s; %d distinct end states, the most common one (crc %08x) on %d robots, robot 0: crc %08x'
          % (os.path.basename(os.environ.get('SOLO_HIP_LIB', 'libsolo_hip.so')), resid, rep, settle_diff, diff, N, stuck, finite, len(rows), crc(common), counts.max(), crc(st[0, :29].view(np.uint32))), flush=True)
    env._close()
