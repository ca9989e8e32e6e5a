"""Averages rocprofv3 --pmc counters per kernel family over the FULL-SIZE dispatches of the
profiled rollout (tools/prof_driver.py) into JSON.

usage: pmc_summary.py out.json pmc_dir [pmc_dir ...]
Families: step = solo_step_kernel<T, true>, outputs = solo_outputs_kernel, returns =
solo_returns_kernel.  Per family only the dispatches with the largest grid are kept (the fused
full-length launches of the timed rollout; the settle loop and the short tail launch have
other names / sizes): per family the dispatches with the grid of the LAST dispatch are kept and the last 6 of them averaged."""
import collections, csv, glob, json, os, sys


def family(name):
  # solo_step_kernel<T, kFull, kResid>: the full-step instantiations (kFull = true), not the physics-only ones
  if 'solo_step_kernel' in name and (', true,' in name or 'Lb1ELb' in name or name.rstrip('>').endswith('true') and name.count(',') == 1):
    return 'step'
  if 'solo_outputs_kernel' in name:
    return 'outputs'
  if 'solo_returns_kernel' in name:
    return 'returns'
  return None


out = collections.defaultdict(dict)
for d in sys.argv[2:]:
  for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    rows = collections.defaultdict(list)  # (family, counter) -> [(grid, value, duration)]
    for r in csv.DictReader(open(f)):
      fam = family(r['Kernel_Name'])
      if fam:
        rows[(fam, r['Counter_Name'])].append((int(r['Grid_Size']), float(r['Counter_Value']),
                                                int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    for (fam, counter), v in rows.items():
      # (the measured launches are the LAST ones of the run - the steady-state preparation in front of them may use
      # another geometry, since round 5 even a larger grid: one slice of 100-step launches)
      gmax = v[-1][0]
      full = [(x, t) for g, x, t in v if g == gmax][-6:]
      out[fam][counter] = sum(x for x, _ in full) / len(full)
      out[fam].setdefault('_dispatches_averaged', len(full))
      out[fam].setdefault('_grid_size', gmax)
      out[fam]['_duration_ns_under_pmc'] = sum(t for _, t in full) / len(full)
json.dump(out, open(sys.argv[1], 'w'), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
