#!/bin/bash
# round-2 profiles at HEAD: kernel stats of the default bench command, PMC traffic passes, PMC passes on the level-0 conv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r2_stats_bench.json 2> $R/gpurun_out/r2_stats_bench.err
echo "stats rc=$?"; ls $R/gpurun_out/r2_stats/*/ | head
cd $R
bash tools/pmc_traffic.sh > gpurun_out/r2_traffic_raw.txt 2>&1; echo "traffic rc=$?"; tail -5 gpurun_out/r2_traffic_raw.txt
bash tools/pmc_conv.sh r2_level0 --level 0 > gpurun_out/r2_pmc_level0.txt 2>&1; echo "pmc0 rc=$?"; tail -30 gpurun_out/r2_pmc_level0.txt
bash tools/pmc_conv.sh r2_level1 --level 1 > gpurun_out/r2_pmc_level1.txt 2>&1; echo "pmc1 rc=$?"
