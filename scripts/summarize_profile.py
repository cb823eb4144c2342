"""Condense a rocprofv3 run (scripts/profile_bench.sh) into small files that can be committed under profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]


def find(sub, pattern):
    hits = glob.glob(os.path.join(out, sub, '**', pattern), recursive=True)
    return hits[0] if hits else None


summary = {}
stats = find('stats', '*kernel_stats.csv')
if stats:
    rows = list(csv.DictReader(open(stats)))
    keep = [{k: r[k] for k in r} for r in rows[:40]]
    with open(os.path.join(out, f'{tag}_kernel_stats.csv'), 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in keep:
            r = dict(r)
            r['Name'] = r['Name'][:160]
            w.writerow(r)
    summary['kernel_stats_top'] = [{'name': r['Name'][:100], 'calls': r.get('Calls'), 'total_ns': r.get('TotalDurationNs'),
                                    'avg_ns': r.get('AverageNs'), 'pct': r.get('Percentage')} for r in rows[:12]]

for sub, counter in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    path = find(sub, '*counter_collection.csv')
    if not path:
        continue
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r.get('Counter_Name') != counter:
            continue
        name = r['Kernel_Name'][:100]
        agg[name][0] += 1
        agg[name][1] += float(r['Counter_Value'])
    top = sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]
    summary[counter] = [{'kernel': k, 'dispatches': v[0], 'sum': v[1], 'per_dispatch': v[1] / max(v[0], 1)} for k, v in top]

with open(os.path.join(out, f'{tag}_summary.json'), 'w') as f:
    json.dump(summary, f, indent=1)
print(json.dumps(summary, indent=1)[:3000])
