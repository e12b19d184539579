export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out; mkdir -p $O
bash tools/prof_round.sh c2 r06 2>&1 | tail -12
bash tools/prof_round.sh steps r06 2>&1 | tail -8
bash tools/prof_round.sh pmc_c2 r06 2>&1 | tail -3
bash tools/prof_round.sh pmc_c3 r06 2>&1 | tail -3
bash tools/prof_round.sh pmc_c5 r06 2>&1 | tail -3
cp $O/r06_c2_pmc_frame.json $O/r06_c3_pmc.json $O/r06_c5_pmc.json $R/profiles/ 2>/dev/null
( time timeout 1500 python bench.py --detail $O/r06_bench_default_detail.json ) > $O/r06_bench_default.json 2> $O/r06_bench_default.err
tail -4 $O/r06_bench_default.err | cut -c1-300
wc -c $O/r06_bench_default.json
python tools/extract_bench.py $O/r06_bench_default.json 2>&1 | head -5
python tools/host_vs_device.py c3 10 fp32 2>&1 | grep -a "HOST_VS\|graph\|eager" 
python tools/host_vs_device.py c3 10 2>&1 | grep -a "HOST_VS\|graph\|eager"
python tools/host_vs_device.py c5 10 fp32 2>&1 | grep -a "HOST_VS\|graph\|eager"
