#!/bin/bash
# rocprofv3 kernel stats of the train workloads; only the small *_kernel_stats.csv files are kept
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
for spec in "supernet_pretrain fp32" "supernet_search bf16" "student_train bf16"; do
  set -- $spec
  out=/tmp/prof_$1_$2
  rm -rf $out
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python $R/bench.py --workload $1 --dtype $2 --steps 3 --warmup 2 > $R/gpurun_out/prof_$1_$2.log 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/r01_$1_$2_kernel_stats.csv
  tail -c 400 $R/gpurun_out/prof_$1_$2.log | grep -o "\"ms_per_step\": [0-9.]*"
done
