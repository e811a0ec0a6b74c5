# usage: prof_cfg.sh <tag> <bench args...>   kernel trace + stats of a short bench run -> gpurun_out/<tag>
set -e
tag=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o f -- python3 $GRAFT_REPO_ROOT/bench.py --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 30 --warmup 10 "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.json 2> $GRAFT_REPO_ROOT/gpurun_out/$tag.err
