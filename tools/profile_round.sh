#!/bin/bash
# Round evidence, on the GPU box from the repo root:  tools/profile_round.sh rNN
#   gpurun_out/<tag>_bench_c2.json        default bench line (roofline + cpu_baseline)
#   gpurun_out/<tag>_stats/               rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>_pmc_{fetch,write}/   separate --pmc passes (one outer iteration, kernels serialised)
set -e
tag=$1
R=$PWD
python bench.py > gpurun_out/${tag}_bench_c2.json 2> gpurun_out/${tag}_bench_c2.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o run -- python3 $R/bench.py --no-cpu-baseline --no-selfcheck > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/${tag}_stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_fetch -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --no-selfcheck > /dev/null 2> $R/gpurun_out/${tag}_pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_write -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --no-selfcheck > /dev/null 2> $R/gpurun_out/${tag}_pmc_write.log
cd $R
# summaries (the raw per-dispatch csv files are tens of MB: only the summaries are kept / merged back)
cp $(find gpurun_out/${tag}_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
python3 tools/pmc_traffic.py $(find gpurun_out/${tag}_pmc_fetch -name "*counter_collection.csv" | head -1) \
    $(find gpurun_out/${tag}_pmc_write -name "*counter_collection.csv" | head -1) \
    gpurun_out/traffic_${tag}.json gpurun_out/${tag}_pmc_traffic_summary.csv
rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write
ls gpurun_out | grep ${tag}
