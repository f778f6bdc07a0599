"""Per-kernel-class wave-cycle breakdown from a rocprofv3 --pmc pass with
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE (the 8 SQ slots of one pass).
usage: sq_summary.py <counter_collection.csv> <out.csv>"""
import collections
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_traffic import klass

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt, dur, seen = collections.Counter(), collections.Counter(), set()
for r in csv.DictReader(open(sys.argv[1])):
    k = klass(r["Kernel_Name"])
    if not k:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); cnt[k] += 1; dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    f.write("kernel_class,launches,total_ms,wave_cycles_parked_pct,wave_cycles_issue_stalled_pct,wave_cycles_issuing_pct,"
            "lds_bank_conflict_pct_of_lds_cycles\n")
    for k, _ in sorted(dur.items(), key=lambda kv: -kv[1])[:14]:
        a = acc[k]; wc = a["SQ_WAVE_CYCLES"] or 1.0
        f.write("%s,%d,%.1f,%.1f,%.1f,%.1f,%.2f\n" % (k, cnt[k], dur[k] / 1e6, 100 * a["SQ_WAIT_ANY"] / wc,
                100 * a["SQ_WAIT_INST_ANY"] / wc, 100 * a["SQ_ACTIVE_INST_ANY"] / wc,
                100 * a["SQ_LDS_BANK_CONFLICT"] / (a["SQ_LDS_IDX_ACTIVE"] or 1.0)))
print(open(sys.argv[2]).read())
