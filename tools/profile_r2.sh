#!/bin/bash
# round-2 profiles at HEAD: kernel stats + trace of the default bench command, PMC traffic passes, PMC passes on the
# level-0 / level-1 conv, the bench line itself
R=$GRAFT_REPO_ROOT
cd $R && python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; echo "bench rc=$?"; tail -2 gpurun_out/r2_bench_final.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r2_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r2_stats_bench.json 2> $R/gpurun_out/r2_stats_bench.err
echo "stats rc=$?"
cd $R
rm -rf gpurun_out/traffic
bash tools/pmc_traffic.sh > gpurun_out/r2_traffic_raw.txt 2>&1; echo "traffic rc=$?"
bash tools/pmc_conv.sh r2_level0 --level 0 > gpurun_out/r2_pmc_level0.txt 2>&1; echo "pmc0 rc=$?"
bash tools/pmc_conv.sh r2_level1 --level 1 > gpurun_out/r2_pmc_level1.txt 2>&1; echo "pmc1 rc=$?"
rm -f gpurun_out/r2_wg_final0.bin gpurun_out/r2_wg_final1.bin
SV_CONV_TRACE=gpurun_out/r2_wg_final0.bin python tools/conv_microbench.py --level 0 --iters 1 > /dev/null 2>&1
SV_CONV_TRACE=gpurun_out/r2_wg_final1.bin python tools/conv_microbench.py --level 1 --iters 1 > /dev/null 2>&1
python tools/wg_trace.py gpurun_out/r2_wg_final0.bin > gpurun_out/r2_wg_final0.txt; python tools/wg_trace.py gpurun_out/r2_wg_final1.bin > gpurun_out/r2_wg_final1.txt
python tools/cfg_timings.py > gpurun_out/r2_cfg_timings.txt 2>&1; tail -12 gpurun_out/r2_cfg_timings.txt
