"""Per-layer conv timing table from `bench.py --dump FILE` (HIP events around every conv launch).  usage: conv_dump.py FILE [filter] [steps]"""
import json, sys
d = json.load(open(sys.argv[1]))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = []
for var, v in d.items():
    for lab, b in v['by_label'].items():
        rows.append((b['ms'] / steps, b['launches'] // steps, var, lab, b['flops'] / (b['ms'] * 1e-3) / 1e12 if b['ms'] > 0 else 0))
rows.sort(reverse=True)
print('total conv ms/step', round(sum(r[0] for r in rows), 3))
for r in rows:
    if flt in r[3] or flt in r[2]:
        print(f'{r[0]:7.3f} ms {r[1]:3d}x {r[4]:7.1f} TF  {r[2]:28s} {r[3]}')
