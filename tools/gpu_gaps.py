"""GPU idle time between kernels from a rocprofv3 kernel trace:  gpu_gaps.py <kernel_trace.csv>
Prints busy time, span and the gap histogram of the busiest 60 % of the trace (the timed steps)."""
import csv
import sys

rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:50]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
n = len(rows)
rows = rows[int(0.35 * n):int(0.95 * n)]          # skip model build / warm-up and the tail
busy = 0; gaps = []; cur_end = rows[0][0]
for s, e, _ in rows:
    if s > cur_end:
        gaps.append(s - cur_end)
    busy += max(0, e - max(s, cur_end)) if e > cur_end else 0
    cur_end = max(cur_end, e)
span = cur_end - rows[0][0]
print(f'kernels {len(rows)}  span {span / 1e6:.1f} ms  busy {busy / 1e6:.1f} ms ({100 * busy / span:.1f} %)  idle {(span - busy) / 1e6:.1f} ms')
import collections
h = collections.Counter()
for g in gaps:
    k = '<1us' if g < 1000 else '<2us' if g < 2000 else '<5us' if g < 5000 else '<10us' if g < 10000 else '<50us' if g < 50000 else '>=50us'
    h[k] += g
for k in ('<1us', '<2us', '<5us', '<10us', '<50us', '>=50us'):
    print(f'  gaps {k:7s}: {h[k] / 1e6:8.2f} ms total')
big = []
cur_end = rows[0][0]; prev = rows[0][2]
for s, e, name in rows:
    if s - cur_end > 30000:
        big.append((s - cur_end, prev, name))
    if e > cur_end:
        cur_end = e; prev = name
print(f'{len(big)} gaps > 30 us; by (kernel before -> kernel after):')
agg = collections.defaultdict(lambda: [0, 0])
for g, a, b in big:
    k = (a.split('(')[0][:34], b.split('(')[0][:34]); agg[k][0] += 1; agg[k][1] += g
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'  {c:4d} x  {t / c / 1e3:7.1f} us avg  {t / 1e6:6.2f} ms   {k[0]}  ->  {k[1]}')
if len(sys.argv) > 2:
    # context of the largest gaps: the kernels around them
    cur_end = rows[0][0]
    idx_big = []
    for i, (s, e, name) in enumerate(rows):
        if s - cur_end > 300000:
            idx_big.append((s - cur_end, i))
        cur_end = max(cur_end, e)
    for g, i in idx_big[:int(sys.argv[2])]:
        print(f'--- gap {g / 1e3:.0f} us before kernel #{i}')
        for k in range(max(0, i - 5), min(len(rows), i + 5)):
            s, e, name = rows[k]
            print(f'   {"*" if k == i else " "} {(e - s) / 1e3:7.1f} us  {name}')
