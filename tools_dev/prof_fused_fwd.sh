# kernel stats of a short configs[2] / configs[3] run with the fused forward stage (BLH_FWD_FUSE=1) and without
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_fusedfwd
mkdir -p $O
for c in 2 3; do
  BLH_FWD_FUSE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fused_cfg$c -o f -- python3 $R/bench.py --config $c --no-configs --no-cpu-baseline --no-alt --steps 30 --warmup 10 > $O/fused_cfg$c.json 2> $O/fused_cfg$c.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/plain_cfg$c -o f -- python3 $R/bench.py --config $c --no-configs --no-cpu-baseline --no-alt --steps 30 --warmup 10 > $O/plain_cfg$c.json 2> $O/plain_cfg$c.err
  for m in fused plain; do
    echo "== configs[$c] $m" >> $O/summary.txt
    python3 $R/tools_dev/prof_summary.py $O/${m}_cfg$c >> $O/summary.txt 2>&1 || true
  done
done
cat $O/summary.txt
