#!/bin/bash
# one rocprofv3 --pmc pass over ONE outer iteration of the benched workload (kernels serialised):
#   [PMC_WORKLOAD=c5] tools/pmc_pass.sh <tag> <counter> [<counter> ...]      -> gpurun_out/<tag>/ + gpurun_out/<tag>.csv
set -e
tag=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$tag -o run -- python3 $R/bench.py --workload ${PMC_WORKLOAD:-c2} --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --no-selfcheck --no-other-workloads > /dev/null 2> $R/gpurun_out/$tag.log
cd $R
f=$(find gpurun_out/$tag -name "*counter_collection.csv" | head -1)
python3 tools/pmc_generic.py $f gpurun_out/$tag.csv > /dev/null
rm -rf gpurun_out/$tag      # the raw per-dispatch csv is tens of MB; the summary is what is kept
cat gpurun_out/$tag.csv
