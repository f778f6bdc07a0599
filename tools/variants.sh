#!/bin/bash
# Run bench.py once per kernel-variant library (built with build_ext --variant TAG ...), serial streams
# so the per-launch HIP-event timings are undisturbed, and dump the per-launch CSVs for agg_launches.py.
# usage: tools/variants.sh TAG [TAG ...]      (on the GPU box, from the repo root)
for tag in "$@"; do
  lib=multimodal_dataset_distillation_amd/variants/libmdd_hip.$tag.so
  MDD_HIP_LIB=$PWD/$lib MDD_SIDE_STREAM=0 python bench.py --steps 2 --warmup 1 \
      --dump-launches gpurun_out/var_$tag.csv > gpurun_out/var_$tag.json 2> gpurun_out/var_$tag.err || exit 1
  MDD_HIP_LIB=$PWD/$lib python bench.py --steps 4 --warmup 2 --no-roofline > gpurun_out/var_${tag}_full.json 2>> gpurun_out/var_$tag.err || exit 1
  python - <<PY
import json
a=json.load(open("gpurun_out/var_$tag.json")); b=json.load(open("gpurun_out/var_${tag}_full.json"))
print("$tag: serial %.1f ms/iter, overlapped %.1f ms/iter" % (a["ms_per_step"], b["ms_per_step"]))
PY
done
