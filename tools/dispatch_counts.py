"""Dispatches per training step by kernel family from a rocprofv3 kernel_stats.csv:  dispatch_counts.py <kernel_stats.csv> <steps>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 7.0
fam = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r['Name'].split('<')[0].split('(')[0].replace('void ', '').strip()
    if name.startswith('_Z'):
        m = re.match(r'_Z(\d+)', name)
        name = name[2 + len(m.group(1)):2 + len(m.group(1)) + int(m.group(1))] if m else name
    fam[name][0] += int(r['Calls']); fam[name][1] += float(r['TotalDurationNs'])
tot = sum(v[0] for v in fam.values())
print(f'{tot / steps:.0f} dispatches per step')
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f'{c / steps:8.1f} /step  {t / c / 1e3:8.1f} us avg  {t / steps / 1e6:7.2f} ms/step  {k[:70]}')
