# usage: bash tools/prof_step.sh c4 5 name   -> gpurun_out/<name>_kernel_stats.csv = per-step kernel table of the timed steps only
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_step
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_step -o run -- python $R/tools/profile_step.py $1 $2 > $R/gpurun_out/prof_step_$3.log 2>&1
TRACE=$(find /tmp/prof_step -name "*kernel_trace.csv" | head -1)
python - <<P
import csv, re
log=open('$R/gpurun_out/prof_step_$3.log').read()
b=int(re.search(r"PROFILE_STEPS_BEGIN \d+ (\d+)", log).group(1)); e=int(re.search(r"PROFILE_STEPS_END \d+ (\d+)", log).group(1)); n=$2
agg={}
with open('$TRACE') as f:
    for r in csv.DictReader(f):
        s=int(r['Start_Timestamp']); 
        if s<b or s>e: continue
        d=int(r['End_Timestamp'])-s
        a=agg.setdefault(r['Kernel_Name'],[0,0]); a[0]+=1; a[1]+=d
rows=sorted(agg.items(), key=lambda kv:-kv[1][1])
tot=sum(v[1] for _,v in rows); calls=sum(v[0] for _,v in rows)
with open('$R/gpurun_out/$3_kernel_stats.csv','w') as f:
    w=csv.writer(f); w.writerow(["Name","CallsPerStep","AverageNs","MsPerStep","Percentage"])
    for k,v in rows: w.writerow([k, v[0]/n, v[1]/v[0], v[1]/1e6/n, 100.0*v[1]/tot])
print("wall ms/step %.2f | kernel ms/step %.2f | launches/step %.0f"%((e-b)/1e6/n, tot/1e6/n, calls/n))
for k,v in rows[:30]:
    print("%-78s calls/step %8.1f avg %8.2f us  ms/step %7.3f  %5.1f%%"%(k[:78], v[0]/n, v[1]/v[0]/1e3, v[1]/1e6/n, 100.0*v[1]/tot))
P
