#!/bin/bash
# round 4, call d: ablations of conv_igemm2 (which part of a stage costs the time?) - kernel durations from the rocprofv3 trace
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for shape in "6 96 96 32 64" "6 192 192 16 32" "6 384 384 8 16" "6 32 32 16 32"; do
  for code in 1100 1120 1121 1122 1104 1123 1124 1125 -2; do
    rm -rf /tmp/kt; timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o run -- python $R/tools/conv_one.py $shape $code bf16 dgrad 100 > /dev/null 2>&1
    f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
    python - <<P
import csv
for r in csv.DictReader(open("$f")):
    if "conv_igemm" in r["Name"]:
        print("$shape code $code: %-70s calls %s avg %.2f us min %.2f us" % (r["Name"][9:79], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
P
  done
done
