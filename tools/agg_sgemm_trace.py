import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"]
    if "sgemm_small" not in n: continue
    mode = "T" if "<1>" in n else "P"
    key = (mode, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r["Queue_Id"])
    a = agg[key]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(list(rows[0].keys()))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(k, a[0], "total %.1f ms avg %.1f us" % (a[1] / 1e3, a[1] / a[0]))
