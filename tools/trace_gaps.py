"""Summarise a rocprofv3 kernel trace (CSV): per-queue busy/idle time, per-kernel totals and the idle
gaps on the busiest queue keyed by (previous kernel -> next kernel).
usage: python tools/trace_gaps.py <kernel_trace.csv> [window_start_ms]"""
import collections
import csv
import re
import sys


def short(n):
    m = re.search(r"k_[a-z0-9_]+", n)
    return m.group(0) if m else n[:32]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    w0 = t0 + (float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0)
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= w0]
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    span = (max(int(r["End_Timestamp"]) for r in rows) - min(int(r["Start_Timestamp"]) for r in rows)) / 1e6
    print(f"window span {span:.1f} ms, {len(rows)} dispatches")
    main_q = max(byq, key=lambda q: sum(e - s for s, e, _ in byq[q]))
    for q, v in byq.items():
        v.sort()
        busy = sum(e - s for s, e, _ in v) / 1e6
        print(f"queue {q}: {len(v)} dispatches, busy {busy:.1f} ms ({100 * busy / span:.0f}%)")
    tot = collections.defaultdict(lambda: [0, 0.0])
    for q, v in byq.items():
        for s, e, n in v:
            tot[(q, n)][0] += 1
            tot[(q, n)][1] += (e - s) / 1e6
    print("--- kernels (queue, name, calls, total ms, avg us)")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"  q{k[0]} {k[1]:30s} {v[0]:6d} {v[1]:9.1f} {1e3 * v[1] / v[0]:8.1f}")
    v = byq[main_q]
    pair = collections.defaultdict(lambda: [0, 0.0])
    for i in range(len(v) - 1):
        g = (v[i + 1][0] - v[i][1]) / 1e3
        if g > 0:
            pair[(v[i][2], v[i + 1][2])][0] += 1
            pair[(v[i][2], v[i + 1][2])][1] += g
    print(f"--- idle gaps on queue {main_q}: total {sum(x[1] for x in pair.values()) / 1e3:.1f} ms")
    for k, x in sorted(pair.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"  {k[0]:26s} -> {k[1]:26s} n {x[0]:5d} sum {x[1] / 1e3:7.1f} ms avg {x[1] / x[0]:6.1f} us")


if __name__ == "__main__":
    main()
