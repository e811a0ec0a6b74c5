# Round-end measurement set (one gpurun call): driver-form bench, long bench, per-config lines,
# rocprofv3 kernel stats + trace of the driver-form command.  Outputs under gpurun_out/final_<tag>/.
set -e
tag=${1:-r03}
out=gpurun_out/final_$tag
mkdir -p $out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err
echo "driver-form bench done"
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "default bench done"
for c in 2 3 4; do
  python3 bench.py --config $c --no-cpu-baseline --no-alt --steps 300 --warmup 100 > $out/bench_cfg$c.json 2> $out/bench_cfg$c.err
done
python3 bench.py --batch 64 --no-configs --no-cpu-baseline --no-alt > $out/bench_cfg0.json 2> $out/bench_cfg0.err
echo "config benches done"
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_driver -o f -- python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $root/$out/prof_driver.json 2> $root/$out/prof_driver.err
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_cfg2 -o f -- python3 $root/bench.py --config 2 --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 30 --warmup 10 > $root/$out/prof_cfg2.json 2> $root/$out/prof_cfg2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_cfg1 -o f -- python3 $root/bench.py --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 30 --warmup 10 > $root/$out/prof_cfg1.json 2> $root/$out/prof_cfg1.err
# (the driver-form trace is tens of MB — gpurun merges at most 64 MiB back: its per-kernel statistics are what is kept)
rm -f $root/$out/prof_driver/f_kernel_trace.csv
echo "profiles done"
