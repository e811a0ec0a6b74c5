#!/bin/bash
# per-kernel times of the encode stage at each shape (one rocprofv3 run per shape): tools_dev/enc_prof.sh <out-prefix>
out=${1:-gpurun_out/enc_prof}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for shape in "4096 1024" "16384 1024" "8192 1024" "16384 2048"; do
  tag=$(echo $shape | tr ' ' 'x')
  rocprofv3 --kernel-trace -d $R/${out}_$tag -o enc -- python3 $R/tools_dev/enc_time.py $shape > /dev/null 2>&1
  echo "== B x W = $tag (first lines: the 4096 x 1024 clock-ramp run is in every trace)" >> $R/${out}.txt
  python3 $R/tools_dev/prof_kernels.py $R/${out}_$tag/enc_results.db enc_ >> $R/${out}.txt
done
cat $R/${out}.txt
