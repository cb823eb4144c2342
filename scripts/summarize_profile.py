"""Condense a rocprofv3 run (scripts/profile_bench.sh) into small files that can be committed under profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else None


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
    ranked = sorted(agg.items(), key=lambda kv: -kv[1][1])
    rows = [{'kernel': k, 'dispatches': v[0], 'sum': v[1], 'per_dispatch': v[1] / max(v[0], 1)} for k, v in ranked]
    summary[counter] = rows[:12]
    summary[counter + '_all'] = rows

# HBM traffic per launch of the conv kernels, keyed by the names bench.py reports.  Units and corrections as
# /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: the counters are in KB; FETCH_SIZE reports half of
# the bytes of 16-B/lane streaming reads (x2); WRITE_SIZE is used as reported (calibrated here on the AdamW kernel: 16 B/param
# read, 16 B/param written incl. the gradient clear -> FETCH x 2 = 6.0 GB, WRITE = 6.0 GB for 375.6 M parameters).
NAMES = {'igemm3w_kernel<false': 'igemm3_kernel<256x256>', 'igemm3dl_kernel': 'igemm3_kernel<256>', 'igemm3h_kernel': 'igemm3_kernel<256,k32>', 'igemm3d_kernel<true, false': 'igemm3_kernel<256>', 'igemm3d_kernel<true, true': 'igemm3_kernel<256,splitk>', 'igemm3_kernel<128': 'igemm3_kernel<128>', 'wgrad3l_kernel': 'wgrad3l_kernel', 'wgrad3_kernel': 'wgrad3_kernel',
         'igemm_kernel<128, 2, 2, false>': 'igemm_kernel<128,generic>', 'wgrad_kernel<128, 128': 'wgrad_kernel<128,128>',
         'adamw_kernel': 'adamw_kernel'}
traffic = {}
for rk, bk in NAMES.items():
    f = next((r for r in summary.get('FETCH_SIZE_all', []) if rk in r['kernel']), None)
    w = next((r for r in summary.get('WRITE_SIZE_all', []) if rk in r['kernel']), None)
    if f and w:
        traffic[bk] = {'bytes_per_launch': round(f['per_dispatch'] * 1024 * 2 + w['per_dispatch'] * 1024),
                       'fetch_bytes': round(f['per_dispatch'] * 1024 * 2), 'write_bytes': round(w['per_dispatch'] * 1024),
                       'dispatches': f['dispatches'],
                       'source': f'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py (same batch), profiles/{tag}_summary.json; '
                                 'FETCH_SIZE x 2 (gfx950 correction), KB -> bytes'}
summary['traffic'] = traffic

# The roofline of bench.py divides by HIP-event time around each launch; rocprofv3's own average for the same kernel must agree.
bur = os.path.join(out, 'bench_under_rocprof.json')
if stats and os.path.exists(bur):
    try:
        line = [l for l in open(bur).read().strip().splitlines() if l.startswith('{')][-1]
        roof = json.loads(line).get('roofline', {})
        rk = next((k for k, v in NAMES.items() if v == roof.get('kernel')), None)
        row = next((r for r in csv.DictReader(open(stats)) if rk and rk in r['Name']), None)
        if row and roof.get('avg_launch_ms'):
            ev_us, rp_us = roof['avg_launch_ms'] * 1e3, float(row['AverageNs']) / 1e3
            summary['duration_agreement'] = {'kernel': roof['kernel'], 'rocprof_symbol': row['Name'][:100], 'rocprof_calls': int(row['Calls']),
                                             'rocprof_avg_us': round(rp_us, 1), 'bench_hip_event_avg_us': round(ev_us, 1),
                                             'ratio_event_over_rocprof': round(ev_us / rp_us, 4)}
    except Exception as e:                                    # the summary is still useful without this cross-check
        summary['duration_agreement'] = {'error': repr(e)}
summary.pop('FETCH_SIZE_all', None)
summary.pop('WRITE_SIZE_all', None)
# what the PMC figures belong to: bench.py attaches `traffic` to a run only when its kernel sources and batch are THESE (else `traffic_stale`)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench
    summary['meta'] = {'csrc_sha16': bench.csrc_sha16(), 'batch': batch, 'command': 'scripts/profile_bench.sh'}
except Exception as e:
    summary['meta'] = {'error': repr(e), 'batch': batch}

with open(os.path.join(out, f'{tag}_summary.json'), 'w') as f:
    json.dump(summary, f, indent=1)
print(json.dumps(summary, indent=1)[:3000])
