"""Identity of the kernel sources a measurement was taken on: one hash over gym_solo_amd/csrc/*.h, *.hip and
include/*.h.  tools/make_pmc_traffic.py stores it next to every counter profile it writes into
profiles/pmc_traffic.json; bench.py recomputes it and refuses to quote a profile taken on other sources
(`roofline.traffic_profile.stale`)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_files():
  files = sorted(glob.glob(os.path.join(ROOT, 'gym_solo_amd', 'csrc', '*.h')) + glob.glob(os.path.join(ROOT, 'gym_solo_amd', 'csrc', '*.hip')) +
                 glob.glob(os.path.join(ROOT, 'include', '*.h')))
  return files + [os.path.join(ROOT, 'gym_solo_amd', 'csrc', 'Makefile')]


def kernel_source_hash():
  h = hashlib.sha256()
  for path in kernel_source_files():
    h.update(os.path.relpath(path, ROOT).encode())
    h.update(b'\0')
    with open(path, 'rb') as f:
      h.update(f.read())
    h.update(b'\0')
  return h.hexdigest()[:16]
