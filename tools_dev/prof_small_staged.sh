set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_small_staged
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/staged -o f -- python3 $R/bench.py --batch 64 --no-configs --no-cpu-baseline --no-alt --steps 300 --warmup 50 > $O/staged.json 2> $O/staged.err
python3 $R/tools_dev/prof_summary.py $O/staged | head -12
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/staged/f_kernel_trace.csv")))
rows=[r for r in rows if "small_" in r["Kernel_Name"] or "clip_adam" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last full step: find the last clip_adam and walk back 12 kernels
idx=[i for i,r in enumerate(rows) if "clip_adam" in r["Kernel_Name"]]
e=idx[-2]; b=idx[-3]+1
t0=int(rows[b]["Start_Timestamp"]); prev=None
for r in rows[b:e+1]:
    s=int(r["Start_Timestamp"]); en=int(r["End_Timestamp"])
    print("%8.1f us  dur %6.1f  gap %5.1f  %s" % ((s-t0)/1e3,(en-s)/1e3,(s-prev)/1e3 if prev else 0.0,r["Kernel_Name"][:40]))
    prev=en
print("step period %.1f us" % ((int(rows[e]["End_Timestamp"])-int(rows[idx[-3]]["End_Timestamp"]))/1e3))
PY
