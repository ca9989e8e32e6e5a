"""profiles/pmc_traffic.json from the PMC summaries of tools/refresh_profiles.sh.

Usage: python tools/make_pmc_traffic.py gpurun_out/<tag>  (reads pmc_hbm_f32.json, pmc_sq_f32.json)
FETCH_SIZE / WRITE_SIZE come from SEPARATE --pmc passes (TCC slots), rocprofv3 reports them in KB.
MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports wide (16 B/lane) streaming reads by 2x and
other widths are uncalibrated; this kernel loads one dword per lane, so the raw value is kept and
the 2x-corrected total is stated next to it."""
import json, os, sys
src = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hbm = json.load(open(os.path.join(src, 'pmc_hbm_f32.json')))
sq = json.load(open(os.path.join(src, 'pmc_sq_f32.json')))
robots, spl = 2048, 100              # tools/prof_driver.py: 4096 robots, 2 stream slices, 100 steps/launch
n = robots * spl
BYTES_PER_ENV_STEP = 385             # bench.py / DESIGN.md §3 (f32)
fetch = hbm['FETCH_SIZE'] * 1024.0
write = hbm['WRITE_SIZE'] * 1024.0
out = {'float32': {
  'hbm_bytes_per_launch': fetch + write,
  'fetch_bytes_per_launch_raw': fetch,
  'write_bytes_per_launch': write,
  'env_steps_per_launch': n,
  'bytes_per_env_step': (fetch + write) / n,
  'algorithmic_bytes_per_launch': BYTES_PER_ENV_STEP * n,
  'valu_insts_per_env_step': sq['SQ_INSTS_VALU'] / n,
  'salu_insts_per_env_step': sq['SQ_INSTS_SALU'] / n,
  'lds_insts_per_env_step': sq['SQ_INSTS_LDS'] / n,
  'how': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (TCC slots), SQ_INSTS_* in a third, '
         'tools/prof_driver.py (bench workload, trajectories recorded, launches of 2048 robots x 100 steps), mean '
         'over the last 10 full launches, KB*1024.  MI355X_MICROARCH.md: FETCH_SIZE under-reports wide 16-B/lane '
         'streams by 2x; this kernel loads one dword per lane (uncalibrated width), so the raw value is reported; '
         'with the 2x correction the total would be %.1f MB.' % ((2 * fetch + write) / 1e6),
  'reading': 'the STEP KERNEL alone (the filter of tools/pmc_summary.py), below the algorithmic %d B/env-step of the whole '
             'path: the fused launch keeps the 128-B state record in LDS for all its steps (no per-step state read), '
             '~%.0f B/env-step read (actions 48 + parameters/snapshot), ~%.0f B/env-step written (the 128-B step record '
             'the output kernels read + done + event byte); the observations (84 B) and rewards (4 B) are written by '
             'solo_outputs_kernel from those records.' % (BYTES_PER_ENV_STEP, fetch / n, write / n)}}
path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
json.dump(out, open(path, 'w'), indent=1)
print(json.dumps(out, indent=1))
