"""Per-kernel-class sums of every counter in a rocprofv3 --pmc counter_collection.csv.
usage: pmc_generic.py <counter_collection.csv> [<out.csv>]"""
import collections
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_traffic import klass

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt, dur, seen, names = collections.Counter(), collections.Counter(), set(), []
for r in csv.DictReader(open(sys.argv[1])):
    k = klass(r["Kernel_Name"])
    if not k:
        continue
    c = r["Counter_Name"]
    if c not in names:
        names.append(c)
    acc[k][c] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); cnt[k] += 1; dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
lines = ["kernel_class,launches,total_ms," + ",".join(names)]
for k, _ in sorted(dur.items(), key=lambda kv: -kv[1])[:16]:
    lines.append("%s,%d,%.2f," % (k, cnt[k], dur[k] / 1e6) + ",".join("%.4g" % acc[k][c] for c in names))
txt = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt)
print(txt)
