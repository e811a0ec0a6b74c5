# kernel stats of a short batch-64 run: the one-launch step (default) and the multi-launch path (BLH_NO_SMALL_STEP=1),
# then the phase timeline of the one-launch kernel (tools/small_step_bench.hip)
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_small_step
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/one -o f -- python3 $R/bench.py --batch 64 --no-configs --no-cpu-baseline --no-alt --steps 300 --warmup 50 > $O/one.json 2> $O/one.err
BLH_NO_SMALL_STEP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/multi -o f -- python3 $R/bench.py --batch 64 --no-configs --no-cpu-baseline --no-alt --steps 300 --warmup 50 > $O/multi.json 2> $O/multi.err
for m in one multi; do
  echo "== batch 64, $m" >> $O/summary.txt
  python3 $R/tools_dev/prof_summary.py $O/$m >> $O/summary.txt 2>&1 || true
done
$R/bilinear_amd/lib/small_step_bench 2 1024 64 200 > $O/timeline.txt 2>&1
$R/bilinear_amd/lib/grid_barrier_bench > $O/barrier.txt 2>&1
cat $O/summary.txt
