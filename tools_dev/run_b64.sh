B="python bench.py --batch 64 --no-configs --no-cpu-baseline --no-alt --no-strong-line"
run() { "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('   ', d['ms_per_step'])"; }
echo eager; run $B
echo "eager r02 knobs"; BLH_LATE_FORK=1 BLH_F32_BWD_EXCL=0 BLH_SIDE_PRIORITY=normal run $B
echo graph; run $B --graph
echo "eager B=256"; run $B --batch 256
echo "graph B=256"; run $B --batch 256 --graph
