# kernel timeline of one fused step at a given batch (fp32, 2 x 1024; further bench.py flags after the batch)
set -e
B=${1:-1024}
shift || true
EXTRA="$@"
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_b$B
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o f -- python3 $R/bench.py --batch $B $EXTRA --no-configs --no-cpu-baseline --no-alt --steps 100 --warmup 30 > $O/bench.json 2> $O/bench.err
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/p/f_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "clip_adam" in r["Kernel_Name"]]
# pick a step in the timed region (between two clip_adams, the 40th)
b=idx[40]+1; e=idx[41]
t0=int(rows[b]["Start_Timestamp"]); prev=int(rows[b-1]["End_Timestamp"])
for r in rows[b:e+1]:
    s=int(r["Start_Timestamp"]); en=int(r["End_Timestamp"])
    n=r["Kernel_Name"].replace("blh::","").replace("void ","")[:58]
    print("q%s %8.1f dur %6.1f gap %6.1f  %s grid %s" % (r["Queue_Id"][-1], (s-t0)/1e3,(en-s)/1e3,(s-prev)/1e3,n,r["Grid_Size_X"]))
    prev=max(prev,en)
print("step %.1f us, %d kernels" % ((int(rows[e]["End_Timestamp"])-int(rows[idx[40]]["End_Timestamp"]))/1e3, e-b+1))
PY
