"""Condense rocprofv3 CSV output into the small summaries committed under profiles/.

  prof_summary.py stats  <prefix>_kernel_stats.csv <prefix>_kernel_trace.csv <out.md> [title] [bench json with dynamic_lds_bytes]
  prof_summary.py pmc    <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <kernel-substr> <grid> [tag]   (commit: profiles/HEAD_COMMIT, written before the snapshot leaves)
  prof_summary.py step   <prefix>_kernel_trace.csv <out.txt.gz> <steps profiled>      the last step's dispatches in launch order
"""
import collections
import csv
import json
import sys


def stats(stats_csv, trace_csv, out, title, bench_json=None):
    # dynamic LDS per kernel family from the profiled run's own JSON line (bench.py: dynamic_lds_bytes); the kernel trace only has the static segment
    dyn = {}
    if bench_json:
        try:
            for line in open(bench_json):
                if line.startswith('{'):
                    dyn = json.loads(line).get('dynamic_lds_bytes', {}) or dyn
        except OSError:
            pass
    rows = list(csv.DictReader(open(stats_csv)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    lines = [f'# {title}', '', f'total kernel time {tot / 1e6:.2f} ms over {sum(int(r["Calls"]) for r in rows)} dispatches', '',
             '## rocprofv3 --kernel-trace --stats (kernel_stats.csv, top 30)', '',
             '| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---|---|---|---|---|---|']
    for r in rows[:30]:
        lines.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | "
                     f"{float(r['MinNs']) / 1e3:.2f} | {float(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")
    fam = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        name = r['Name']
        name = name.split('<')[0].split('(')[0].replace('void ', '').strip()
        if name.startswith('_Z'):                      # mangled template instantiation: keep the identifier only
            import re
            m = re.match(r'_Z(\d+)', name)
            name = name[2 + len(m.group(1)):2 + len(m.group(1)) + int(m.group(1))] if m else name
        fam[name][0] += int(r['Calls']); fam[name][1] += float(r['TotalDurationNs'])
    lines += ['', '## every kernel family (template arguments folded), by total time', '', '| kernel | calls | total ms | % |', '|---|---|---|---|']
    for k, a in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        lines.append(f'| `{k[:70]}` | {a[0]} | {a[1] / 1e6:.3f} | {100 * a[1] / tot:.2f} |')
    agg = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for r in csv.DictReader(open(trace_csv)):
        name = r['Kernel_Name'].split('(')[0][:60]
        wgs = 1
        for ax in 'XYZ':                               # grid sizes are in work-items per axis: workgroups = product over the three axes
            wgs *= max(1, int(r.get(f'Grid_Size_{ax}', 1) or 1)) // max(1, int(r.get(f'Workgroup_Size_{ax}', 1) or 1))
        fam_ = name.replace('void ', '').split('<')[0].strip()
        m_ = __import__('re').match(r'_Z(\d+)', fam_)
        if m_:
            fam_ = fam_[2 + len(m_.group(1)):2 + len(m_.group(1)) + int(m_.group(1))]
        lds_ = f"{r['LDS_Block_Size']} + {dyn[fam_]} dyn" if fam_ in dyn else r['LDS_Block_Size']
        key = (name, wgs, lds_, r['VGPR_Count'])
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        a = agg[key]; a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    lines += ['', '## per (kernel, workgroups) from kernel_trace.csv (top 40 by total time)', '',
              '| kernel | workgroups | LDS B (static + dynamic at launch) | VGPR | calls | total ms | avg us | min us | max us |', '|---|---|---|---|---|---|---|---|---|']
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        lines.append(f'| `{k[0]}` | {k[1]} | {k[2]} | {k[3]} | {a[0]} | {a[1] / 1e3:.3f} | {a[1] / a[0]:.2f} | {a[2]:.2f} | {a[3]:.2f} |')
    open(out, 'w').write('\n'.join(lines) + '\n')


def pmc(fetch_csv, write_csv, out, substr, grid, tag=''):
    def mean(path, counter):
        v = [float(r['Counter_Value']) for r in csv.DictReader(open(path))
             if r['Counter_Name'] == counter and substr in r['Kernel_Name'] and (not grid or r['Grid_Size'] == grid)]
        return (sum(v) / len(v), len(v)) if v else (None, 0)
    f, nf = mean(fetch_csv, 'FETCH_SIZE')
    w, nw = mean(write_csv, 'WRITE_SIZE')
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half
    # of the bytes of a wide coalesced streaming read -> doubled; WRITE_SIZE is exact for 16-B/lane and dword stores.
    import os
    commit = ''
    try:
        commit = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'HEAD_COMMIT')).read().strip()
    except OSError:
        pass
    res = {'tag': tag or 'untagged', 'commit': commit or 'n/a', 'kernel': substr, 'grid_size': grid, 'launches_fetch': nf, 'launches_write': nw,
           'FETCH_SIZE_KiB_per_launch': f, 'WRITE_SIZE_KiB_per_launch': w,
           'hbm_read_bytes_per_launch': None if f is None else 2 * f * 1024,
           'hbm_write_bytes_per_launch': None if w is None else w * 1024,
           'hbm_bytes_per_launch': None if (f is None or w is None) else 2 * f * 1024 + w * 1024,
           'correction': 'read = 2 x FETCH_SIZE x 1024 (gfx950 half-count), write = WRITE_SIZE x 1024'}
    json.dump(res, open(out, 'w'), indent=1)
    print(res)


def step(trace_csv, out, steps):
    """one line per dispatch of the LAST profiled step (total dispatches / steps of them), in start order: gap to the previous kernel's end, duration,
    workgroups, kernel -- the per-call view the per-family tables fold away (which call of a family is the slow one, where the queue ran dry)"""
    import gzip
    rows = sorted(csv.DictReader(open(trace_csv)), key=lambda r: int(r['Start_Timestamp']))
    n = len(rows) // steps
    rows = rows[-n:]
    prev_end = int(rows[0]['Start_Timestamp'])
    t0 = prev_end
    with gzip.open(out, 'wt') as f:
        f.write(f'# last of {steps} profiled steps: {n} dispatches; columns: start us (from the first), gap us, duration us, workgroups, kernel\n')
        for r in rows:
            st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            wgs = 1
            for ax in 'XYZ':
                wgs *= max(1, int(r.get(f'Grid_Size_{ax}', 1) or 1)) // max(1, int(r.get(f'Workgroup_Size_{ax}', 1) or 1))
            f.write(f"{(st - t0) / 1e3:10.1f} {(st - prev_end) / 1e3:8.1f} {(en - st) / 1e3:9.1f} {wgs:7d} {r['Kernel_Name'].split('(')[0][:70]}\n")
            prev_end = max(prev_end, en)


if __name__ == '__main__':
    if sys.argv[1] == 'step':
        step(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    elif sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else 'rocprofv3 summary', sys.argv[6] if len(sys.argv) > 6 else None)
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6] if len(sys.argv) > 6 else '', sys.argv[7] if len(sys.argv) > 7 else '')
