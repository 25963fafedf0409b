#!/usr/bin/env python3
"""Turn the raw rocprofv3 output that tools/collect_pmc.sh left under gpurun_out/<tag>/ into the
committed summaries profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc.json (run in the build
container: gpurun only brings gpurun_out/ back).  usage: tools/materialise_profiles.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, 'gpurun_out', tag)
prof = os.path.join(root, 'profiles')
stats = sorted(glob.glob(out + '/stats/*/*kernel_stats.csv'), key=os.path.getmtime)       # the newest run's (file names are pids)
if stats:
    shutil.copy(stats[-1], os.path.join(prof, '%s_kernel_stats.csv' % tag))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pmc_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
for k, d in res.items():
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        # rocprofv3 units: KiB.  gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads
        # (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE is exact.
        d['hbm_bytes_corrected'] = (2.0 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024.0
sys.path.insert(0, root)
import bench                                                   # csrc_hash(): which kernel sources these counters belong to
# gpurun ships the working tree as it is, so the sources profiled are the ones in the tree when the collection ran: the script
# that collects (tools/collect_pmc.sh, collect_eval_pmc.sh) records `csrc_hash` next to the counters; fall back to the tree's
stamp = None
hfile = os.path.join(out, 'csrc_hash.txt')
if os.path.exists(hfile):
    stamp = open(hfile).read().strip()
json.dump({'tag': tag, 'note': 'averages per dispatch; FETCH_SIZE doubled per the gfx950 correction', 'csrc_hash': stamp or bench.csrc_hash(),
           'csrc_hash_source': 'recorded on the GPU box by the collection script' if stamp else 'working tree at materialisation time', 'kernels': res},
          open(os.path.join(prof, '%s_pmc.json' % tag), 'w'), indent=1, sort_keys=True)
for name in ('bench_line_unprofiled.json', 'clock_probe.txt'):
    src = os.path.join(out, name)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(prof, '%s_%s' % (tag, name.replace('_unprofiled', ''))))
log = os.path.join(out, 'bench_stats.log')
if os.path.exists(log):
    lines = [l for l in open(log) if l.startswith('{')]
    if lines:
        open(os.path.join(prof, '%s_bench_line_under_rocprof.json' % tag), 'w').write(lines[-1])
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get('hbm_bytes_corrected', 0))[:8]:
    hit = d.get('TCC_HIT_sum', 0) / max(1.0, d.get('TCC_HIT_sum', 0) + d.get('TCC_MISS_sum', 0))
    print('%-60s %7.1f MB  L2 hit %.2f' % (k[:60], d.get('hbm_bytes_corrected', 0) / 1e6, hit))
