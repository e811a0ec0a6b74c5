# one-step kernel timeline of a data-parallel run: tools_dev/prof_dp_tag.sh <tag> <bench args...>  -> gpurun_out/<tag>_timeline.txt
set -e
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${tag}_prof
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/p -o f -- python3 $R/bench.py --gpus 1 --rehearse-rccl "$@" --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 60 --warmup 20 > $O/bench.json 2> $O/bench.err
python3 - > $R/gpurun_out/${tag}_timeline.txt <<PY
import csv
rows=list(csv.DictReader(open("$O/p/f_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "clip_adam" in r["Kernel_Name"]]
b=idx[-20]+1; e=idx[-19]
t0=int(rows[b]["Start_Timestamp"]); prev=int(rows[b-1]["End_Timestamp"])
for r in rows[b:e+1]:
    s=int(r["Start_Timestamp"]); en=int(r["End_Timestamp"])
    n=r["Kernel_Name"].replace("blh::","").replace("void ","")[:60]
    print("q%s %8.1f dur %6.1f gap %6.1f  %s grid %s" % (r["Queue_Id"][-1], (s-t0)/1e3,(en-s)/1e3,(s-prev)/1e3,n,r["Grid_Size_X"]))
    prev=max(prev,en)
print("step %.1f us, %d kernels" % ((int(rows[e]["End_Timestamp"])-int(rows[idx[-20]]["End_Timestamp"]))/1e3, e-b+1))
PY
rm -rf $O/p
