"""Aggregate a bench.py --dump-launches CSV: total ms per contraction kind and the heaviest layer shapes."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
kinds = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
agg = collections.OrderedDict()
for r in rows:
    k = kinds[r["kind"]]
    k[0] += 1; k[1] += float(r["ms"]); k[2] += float(r["gflops"]); k[3] += float(r["gbytes"])
    key = tuple(r[c] for c in ("kind", "transposed", "nsrc", "M", "nc", "groups", "kc", "k", "stride"))
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += float(r["ms"]); a[2] += float(r["gflops"]); a[3] += float(r["gbytes"])
print("total ms %.2f over %d launches" % (sum(k[1] for k in kinds.values()), len(rows)))
for kk, k in sorted(kinds.items()):
    print("kind %s: n %d  %.2f ms  %.0f TF/s  %.0f GB/s" % (kk, k[0], k[1], k[2] / k[1], k[3] / k[1] * 1e3))
print("kind tr ns M nc g kc k s | n ms avg_us TF/s GB/s")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(" ".join("%7s" % x for x in key), "|", a[0], "%.2f %.1f %.0f %.0f" % (a[1], a[1] / a[0] * 1e3, a[2] / a[1], a[3] / a[1] * 1e3))
