#!/bin/bash
# round-4 profiles at HEAD.  Each step writes under gpurun_out/ (nothing is silent for long).  Counters in their own passes.
R=$GRAFT_REPO_ROOT
cd $R && python bench.py > gpurun_out/r4_bench_final.json 2> gpurun_out/r4_bench_final.err; echo "bench rc=$?"; tail -2 gpurun_out/r4_bench_final.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r4_stats $R/gpurun_out/r4_cfg5_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_stats -- python3 $R/bench.py --no-cpu-baseline --strong-frames 64 > $R/gpurun_out/r4_stats_bench.json 2> $R/gpurun_out/r4_stats_bench.err
echo "stats rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_cfg5_stats -- python3 $R/tools/cfg5_profile.py 8 > $R/gpurun_out/r4_cfg5_profile.txt 2>&1
echo "cfg5 stats rc=$?"; tail -1 $R/gpurun_out/r4_cfg5_profile.txt
cd $R
python tools/cfg5_profile.py 10 > gpurun_out/r4_cfg5_plain.txt 2>&1; tail -1 gpurun_out/r4_cfg5_plain.txt
for f in gpurun_out/r4_stats/*/*kernel_trace.csv; do python tools/kernel_by_grid.py $f > gpurun_out/r4_bench_kernel_by_grid.txt 2>&1; done
for f in gpurun_out/r4_cfg5_stats/*/*kernel_trace.csv; do python tools/kernel_by_grid.py $f > gpurun_out/r4_cfg5_kernel_by_grid.txt 2>&1; done
rm -rf gpurun_out/traffic
bash tools/pmc_traffic.sh > gpurun_out/r4_traffic_raw.txt 2>&1; echo "traffic rc=$?"
bash tools/pmc_conv.sh r4_level0 --level 0 --split 9,18 > gpurun_out/r4_pmc_level0.txt 2>&1; echo "pmc0 rc=$?"
bash tools/pmc_conv.sh r4_level1 --level 1 --split 9,18 > gpurun_out/r4_pmc_level1.txt 2>&1; echo "pmc1 rc=$?"
bash tools/pmc_conv.sh r4_cfg5_level0 --points 500000 --scale 100 --level 0 --split 9,18 > gpurun_out/r4_pmc_cfg5_level0.txt 2>&1; echo "pmc cfg5 rc=$?"
bash tools/pmc_kernel.sh r4_thin_l0 conv_thin_lds --cin 32 --cout 32 --level 0 > gpurun_out/r4_pmc_thin_lds_level0.txt 2>&1; echo "pmc thin rc=$?"
python tools/engine_stream_phases.py 1 3 > gpurun_out/r4_engine_stream.txt 2>&1; tail -4 gpurun_out/r4_engine_stream.txt
python tools/predict_stream_phases.py 4 > gpurun_out/r4_predict_stream_phases.txt 2>&1; tail -2 gpurun_out/r4_predict_stream_phases.txt
python tools/hbm_layers_microbench.py 0 > gpurun_out/r4_hbm_layers.txt 2>&1; tail -8 gpurun_out/r4_hbm_layers.txt
