"""Averages rocprofv3 --pmc counters of solo_step_kernel dispatches (last 100) into JSON."""
import csv, glob, json, os, sys, collections
out = {}
for d in sys.argv[2:]:
  for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
      if 'solo_step_kernel' in r['Kernel_Name'] and ('true' in r['Kernel_Name'] or 'Lb1' in r['Kernel_Name']):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
      v = v[-10:]
      out[k] = sum(v) / len(v)
json.dump(out, open(sys.argv[1], 'w'), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
