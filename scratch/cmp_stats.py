"""compare two profiles/summarize.py `stats` summaries by kernel family:  python scratch/cmp_stats.py a.csv b.csv [steps]"""
import csv, re, sys
def load(p):
    d = {}
    for r in csv.DictReader(l for l in open(p) if not l.startswith('#')):
        x = d.setdefault(re.sub(r'<.*', '', r['kernel']) if len(sys.argv) < 5 else r['kernel'], [0, 0.0]); x[0] += int(r['calls']); x[1] += float(r['total_ms'])
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
n = float(sys.argv[3]) if len(sys.argv) > 3 else 7
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, [0, 0])[1] + b.get(k, [0, 0])[1])):
    x, y = a.get(k, [0, 0]), b.get(k, [0, 0])
    if max(x[1], y[1]) / n > 0.02:
        print(f"{k[:90]:90s} {x[0]:6d} {x[1] / n:8.3f}   {y[0]:6d} {y[1] / n:8.3f}   d={(x[1] - y[1]) / n:+.3f}")
print("sum", sum(v[1] for v in a.values()) / n, sum(v[1] for v in b.values()) / n)
