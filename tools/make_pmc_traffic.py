"""profiles/pmc_traffic.json from the PMC summaries of tools/refresh_profiles.sh.

usage: python tools/make_pmc_traffic.py gpurun_out/<tag> <config>   (config = f32 | f32_k20 | f64 | f64_k20:
reads pmc_hbm_<config>.json, pmc_sq_<config>.json, prof_driver_<config>.json; MERGES the entry
"float32" / "float32_k20" / "float64" / "float64_k20" into profiles/pmc_traffic.json)

Everything is normalised PER ENV-STEP, so that bench.py can scale it to the launch size of its own
run (roofline.traffic = hbm_bytes_per_env_step x env_steps_per_launch of that run).  FETCH_SIZE /
WRITE_SIZE come from SEPARATE --pmc passes (TCC slots), rocprofv3 reports them in KB.
MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE under-reports WIDE (16 B/lane) streaming
reads by exactly 2x, WRITE_SIZE is exact for 16 B/lane stores, other widths are uncalibrated.  The
step kernel loads one dword per lane (uncalibrated: raw value kept); its output epilogue (round 3; before: the
outputs kernel) re-reads its own records and writes the rows with dword stores.  The raw figures are reported, with the
2x-on-reads upper bound next to them."""
import json, os, sys
src, cfg = sys.argv[1], (sys.argv[2:] or ['f32'])[0]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_solo_amd.build_info import kernel_source_hash
hbm = json.load(open(os.path.join(src, 'pmc_hbm_%s.json' % cfg)))
sq = json.load(open(os.path.join(src, 'pmc_sq_%s.json' % cfg)))
meta = json.load(open(os.path.join(src, 'prof_driver_%s.json' % cfg)))   # robots per launch, steps per launch
n = meta['robots_per_launch'] * meta['steps_per_launch']       # env-steps one launch (of each kernel) covers
BYTES_PER_ENV_STEP = {'float32': 385, 'float64': 765}[meta['dtype']]   # bench.py / SURVEY.md §8d
key = meta['dtype'] + ('_k20' if cfg.endswith('_k20') else '')
per = {}
for fam in ('step', 'outputs', 'returns'):
  h = hbm.get(fam, {})
  per[fam] = {'read_bytes_per_env_step': h.get('FETCH_SIZE', 0.0) * 1024.0 / n,
              'write_bytes_per_env_step': h.get('WRITE_SIZE', 0.0) * 1024.0 / n}
rd = sum(p['read_bytes_per_env_step'] for p in per.values())
wr = sum(p['write_bytes_per_env_step'] for p in per.values())
s = sq['step']
wave_cycles = s.get('SQ_WAVE_CYCLES')
out = {key: {
  # the kernel sources the counters were collected on (bench.py drops the profile when they have changed since)
  'kernel_source_hash': kernel_source_hash(),
  'migrate_steps': meta.get('migrate_steps', 0),
  'env_steps_per_profiled_launch': n,
  'steps_per_launch': meta['steps_per_launch'], 'robots_per_launch': meta['robots_per_launch'], 'launch_chains': meta.get('launch_chains', 1),
  'hbm_bytes_per_env_step': rd + wr,
  'hbm_read_bytes_per_env_step': rd,
  'hbm_write_bytes_per_env_step': wr,
  'hbm_bytes_per_env_step_reads_doubled': 2 * rd + wr,
  'per_kernel': per,
  'algorithmic_bytes_per_env_step': BYTES_PER_ENV_STEP,
  'valu_insts_per_env_step': s['SQ_INSTS_VALU'] / n,
  'insts_per_env_step': (s['SQ_INSTS'] / n) if 'SQ_INSTS' in s else None,
  'branch_insts_per_env_step': (s['SQ_INSTS_BRANCH'] / n) if 'SQ_INSTS_BRANCH' in s else None,
  'salu_insts_per_env_step': s['SQ_INSTS_SALU'] / n,
  'lds_insts_per_env_step': s['SQ_INSTS_LDS'] / n,
  'smem_insts_per_env_step': s.get('SQ_INSTS_SMEM', 0.0) / n,
  'step_kernel_wave_cycle_shares': None if not wave_cycles else {
    k: s[c] / wave_cycles for k, c in (('wait_any_parked_at_waitcnt', 'SQ_WAIT_ANY'), ('wait_inst_any_issue_stall', 'SQ_WAIT_INST_ANY'),
                                       ('active_inst_any', 'SQ_ACTIVE_INST_ANY'), ('active_inst_valu', 'SQ_ACTIVE_INST_VALU'),
                                       ('active_inst_sca', 'SQ_ACTIVE_INST_SCA'), ('active_inst_lds', 'SQ_ACTIVE_INST_LDS'),
                                       ('wait_inst_lds', 'SQ_WAIT_INST_LDS')) if c in s},
  # the SIMDs' VALU busy share over the launch, MEASURED: SQ_ACTIVE_INST_VALU (quad-cycles, summed over the waves) x 4 /
  # (1024 SIMDs x the launch's cycles = GRBM_GUI_ACTIVE / 8 XCDs) - bench.py's roofline.secondary.frac
  'valu_busy_share_measured': (s['SQ_ACTIVE_INST_VALU'] * 4.0 / (1024 * s['GRBM_GUI_ACTIVE'] / 8.0)) if ('SQ_ACTIVE_INST_VALU' in s and s.get('GRBM_GUI_ACTIVE')) else None,
  'step_kernel_busy_cycles': s.get('SQ_BUSY_CYCLES'), 'step_kernel_wave_cycles': wave_cycles, 'step_kernel_waves': s.get('SQ_WAVES'),
  'grbm_gui_active_per_launch': s.get('GRBM_GUI_ACTIVE'),
  # MI355X_MICROARCH.md (DVFS): effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time (same pass)
  'effective_clock_hz': (s['GRBM_GUI_ACTIVE'] / 8.0 / (s['_duration_ns_under_pmc'] * 1e-9)) if s.get('GRBM_GUI_ACTIVE') else None,
  'step_kernel_ms_under_pmc_serialised': s.get('_duration_ns_under_pmc', 0) * 1e-6,
  'traffic_note': 'measured HBM bytes per env-step (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, '
                  'KB x 1024, every solo kernel of one fused launch of %d robots x %d steps - since round 3 the step kernel alone, its output epilogue included) x this run\'s '
                  'env-steps per launch; raw FETCH_SIZE (dword-per-lane loads: uncalibrated width; with the guide\'s 2x '
                  'correction for wide reads the total would be %.0f B/env-step); algorithmic figure of the whole path: %d '
                  'B/env-step (a fused launch neither re-reads nor re-writes the state record per step; it writes one 128-B record '
                  'per robot-step, which its output epilogue reads back - L2-resident - to write the observation / reward / done rows)' % (meta['robots_per_launch'], meta['steps_per_launch'], 2 * rd + wr, BYTES_PER_ENV_STEP),
  'how': 'tools/refresh_profiles.sh: tools/prof_driver.py (bench workload, every step recorded) under rocprofv3 --pmc, one '
         'pass per counter group; tools/pmc_summary.py averages the last 6 full-size dispatches per kernel'}}
path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
try:
  table = json.load(open(path))
except Exception:
  table = {}
table.update(out)
json.dump(table, open(path, 'w'), indent=1)
print(json.dumps(out, indent=1))
