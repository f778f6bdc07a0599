"""Per-launch HBM/fabric traffic of the contraction kernels from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE, each in its own run as MI355X_MICROARCH.md prescribes).
traffic/launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes: both counters are in KiB; the factor 2 on
FETCH_SIZE is the guide's gfx950 correction, re-checked here on k_sqdist (reads 2 x P_img x 4 B).
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <out.csv>"""
import collections
import csv
import json
import re
import sys


def klass(name):
    if "k_conv_wgrad" in name:
        return "conv_wgrad"
    if "k_conv_gemm" in name:
        li = re.findall(r"Li(\d+)E", name)
        if len(li) >= 4:
            wgm, wgn, tm, tn = map(int, li[:4])
            return "conv_gemm<%dx%d>" % (wgm * tm * 32, wgn * tn * 32)
        return "conv_gemm"
    m = re.search(r"k_[a-z0-9_]+", name)
    return m.group(0) if m else None


def collect(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = klass(r["Kernel_Name"])
        if k is None:
            continue
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    return acc


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    out, rows = {}, []
    for k in sorted(fetch, key=lambda k: -fetch[k][1]):
        n = fetch[k][0]
        f = 2.0 * fetch[k][1] * 1024 / n
        w = write[k][1] * 1024 / max(1, write[k][0]) if k in write else 0.0
        out[k] = f + w
        rows.append((k, n, int(f), int(w)))
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    with open(sys.argv[4], "w") as fo:
        fo.write("kernel_class,launches,fetch_bytes_x2_per_launch,write_bytes_per_launch\n")
        for r in rows:
            fo.write("%s,%d,%d,%d\n" % r)
    tot = sum((fetch[k][1] * 2 + (write[k][1] if k in write else 0)) * 1024 for k in fetch)
    print("total traffic over the profiled run: %.1f GB" % (tot / 1e9))


if __name__ == "__main__":
    main()
