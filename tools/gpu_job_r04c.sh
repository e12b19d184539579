#!/bin/bash
# round 4, call c: hardware counters of conv_igemm2 on two geometries (what bounds the K loop?)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp
rocprofv3 -L > $O/r04c_counters.txt 2>&1
grep -c . $O/r04c_counters.txt
run() {  # name, pmc list, args...
  name=$1; pmc=$2; shift 2
  rm -rf /tmp/pmc_$name
  timeout 120 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/pmc_$name -o run -- python $R/tools/conv_one.py "$@" > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  python - <<P
import csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
try:
    for r in csv.DictReader(open("$f")):
        k=r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k,c in agg.items():
        if "igemm" in k:
            print("$name", k, "dispatches", len(n[k]), {a: round(v/len(n[k])) for a,v in c.items()})
except Exception as e:
    print("$name failed", e, open("/tmp/pmc_$name.log").read()[-400:])
P
}
for shape in "6 96 96 32 64 1100" "6 96 96 32 64 -2" "6 384 384 8 16 1104" "6 384 384 8 16 -2" "6 192 192 16 32 1100"; do
  tag=$(echo $shape | tr ' ' '_' | tr -d '-')
  run a_$tag "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" $shape
  run b_$tag "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" $shape
  run c_$tag "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum" $shape
done
rm -rf /tmp/kt; timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o run -- python $R/tools/conv_one.py 6 96 96 32 64 1100 > /dev/null 2>&1; f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -5 $f | cut -c1-200
