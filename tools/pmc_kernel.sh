#!/bin/bash
# PMC passes over the single-layer microbenchmark for ANY kernel-name filter (separate passes; no tracing flags mixed in).
#   tools/pmc_kernel.sh <tag> <kernel name substring> <conv_microbench args ...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$1; shift
FILTER=$1; shift
mkdir -p $OUT
python tools/conv_microbench.py "$@" > $OUT/plain.log 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum" ; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python tools/conv_microbench.py "$@" --iters 3 > $OUT/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$FILTER" not in r["Kernel_Name"]: continue
        agg[r["Kernel_Name"][:48]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(r["Kernel_Name"][:48], r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:40s} {v / cnt[(k, c)]:18.1f} per launch ({cnt[(k,c)]} launches)")
PY
cat $OUT/plain.log
