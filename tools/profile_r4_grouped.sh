#!/bin/bash
# round-4 profiles after the pipeline groups 4 frames per sparse tensor (bench.py --group 4 = default).  Counters in their own passes.
R=$GRAFT_REPO_ROOT
cd $R && python bench.py > gpurun_out/r4g_bench.json 2> gpurun_out/r4g_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r4g_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r4g_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4g_stats -- python3 $R/bench.py --no-cpu-baseline --strong-frames 64 > $R/gpurun_out/r4g_stats_bench.json 2> $R/gpurun_out/r4g_stats_bench.err
echo "stats rc=$?"
cd $R
for f in gpurun_out/r4g_stats/*/*kernel_trace.csv; do python tools/kernel_by_grid.py $f > gpurun_out/r4g_bench_kernel_by_grid.txt 2>&1; done
cp gpurun_out/r4g_stats/*/*kernel_stats.csv gpurun_out/r4g_bench_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/r4g_stats/*/*kernel_trace.csv gpurun_out/traffic
bash tools/pmc_traffic.sh > gpurun_out/r4g_traffic_raw.txt 2>&1; echo "traffic rc=$?"
bash tools/pmc_conv.sh r4g_level0 --level 0 --split 9,18 --frames 4 > gpurun_out/r4g_pmc_level0.txt 2>&1; echo "pmc0 rc=$?"
tail -2 gpurun_out/r4g_pmc_level0.txt
