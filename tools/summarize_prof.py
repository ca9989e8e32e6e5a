"""Trim a rocprofv3 --kernel-trace --stats output directory into a small, committable summary."""
import csv, glob, json, os, statistics, sys

def main(prof_dir, out_path, label):
  stats = glob.glob(os.path.join(prof_dir, '**', '*kernel_stats.csv'), recursive=True)[0]
  trace = glob.glob(os.path.join(prof_dir, '**', '*kernel_trace.csv'), recursive=True)[0]
  rows = list(csv.DictReader(open(stats)))
  out = {'label': label, 'source': 'rocprofv3 --kernel-trace --stats', 'kernel_stats_top': []}
  for r in rows[:6]:
    out['kernel_stats_top'].append({'name': r['Name'][:110], 'calls': int(r['Calls']),
                                    'total_ns': int(r['TotalDurationNs']), 'avg_ns': float(r['AverageNs']),
                                    'pct': float(r['Percentage']), 'min_ns': int(r['MinNs']), 'max_ns': int(r['MaxNs'])})
  # the kernel of the timed workload = the step-kernel instantiation with the largest total time
  step_rows = [r for r in rows if 'solo_step_kernel' in r['Name']]
  dominant = max(step_rows, key=lambda r: int(r['TotalDurationNs']))['Name'] if step_rows else ''
  out['dominant_step_kernel'] = dominant
  tr = [r for r in csv.DictReader(open(trace)) if r['Kernel_Name'] == dominant]
  # only full-size fused launches (the timed workload): the largest grid and the modal duration class
  if tr:
    gmax = max(int(r['Grid_Size_X']) for r in tr)
    tr = [r for r in tr if int(r['Grid_Size_X']) == gmax]
  if tr:
    # dispatches of the timed workload only: the largest common grid
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in tr]
    # in dispatch order: the steady-state preparation (1000 untimed steps from 4096 synchronised episodes), warm-up, the
    # untimed repeat, the timed repeats, the roofline's launches
    out['solo_step_kernel_durations_ns_in_dispatch_order'] = d
    tail = d[-36:] if len(d) >= 36 else d   # (the untimed repeat + 30 timed repeats + 5 roofline launches of the driver's command)
    out['solo_step_kernel_last_36'] = {'avg_ns': statistics.mean(tail), 'median_ns': statistics.median(tail), 'min_ns': min(tail), 'max_ns': max(tail)}
    out['solo_step_kernel'] = {'dispatches': len(d), 'avg_ns': statistics.mean(d), 'median_ns': statistics.median(d),
                               'min_ns': min(d), 'max_ns': max(d), 'grid_x': tr[-1]['Grid_Size_X'],
                               'workgroup_x': tr[-1]['Workgroup_Size_X'], 'vgpr': tr[-1]['VGPR_Count'],
                               'agpr': tr[-1]['Accum_VGPR_Count'], 'sgpr': tr[-1]['SGPR_Count'],
                               'lds_bytes': tr[-1]['LDS_Block_Size'], 'scratch': tr[-1]['Scratch_Size']}
  json.dump(out, open(out_path, 'w'), indent=1)
  print(json.dumps(out.get('solo_step_kernel'), indent=1))

if __name__ == '__main__':
  main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else '')
