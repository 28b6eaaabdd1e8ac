#!/bin/bash
# round-3 profiles at HEAD: the bench line, kernel stats + trace of the default bench command, PMC traffic passes, PMC
# passes on the level-0 / level-1 conv, engine stages and streaming phases, configuration timings
R=$GRAFT_REPO_ROOT
cd $R && python bench.py > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err; echo "bench rc=$?"; tail -2 gpurun_out/r3_bench_final.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r3_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_stats -- python3 $R/bench.py --no-cpu-baseline --strong-frames 64 > $R/gpurun_out/r3_stats_bench.json 2> $R/gpurun_out/r3_stats_bench.err
echo "stats rc=$?"
cd $R
rm -rf gpurun_out/traffic
bash tools/pmc_traffic.sh > gpurun_out/r3_traffic_raw.txt 2>&1; echo "traffic rc=$?"
# the dominant layer as the frame runs it: three offset-range passes (and the single launch beside it)
bash tools/pmc_conv.sh r3_level0 --level 0 --split 9,18 > gpurun_out/r3_pmc_level0.txt 2>&1; echo "pmc0 rc=$?"
bash tools/pmc_conv.sh r3_level0_single --level 0 > gpurun_out/r3_pmc_level0_single.txt 2>&1; echo "pmc0s rc=$?"
bash tools/pmc_conv.sh r3_level1 --level 1 --split 9,18 > gpurun_out/r3_pmc_level1.txt 2>&1; echo "pmc1 rc=$?"
python tools/engine_stages.py > gpurun_out/r3_engine_stages.txt 2>&1
python tools/engine_stream_phases.py > gpurun_out/r3_engine_stream.txt 2>&1; tail -5 gpurun_out/r3_engine_stream.txt
python tools/cfg_timings.py > gpurun_out/r3_cfg_timings.txt 2>&1; tail -8 gpurun_out/r3_cfg_timings.txt
python tools/cfg3_fullsize_timing.py 64 > gpurun_out/r3_cfg3_fullsize.txt 2>&1; tail -3 gpurun_out/r3_cfg3_fullsize.txt
