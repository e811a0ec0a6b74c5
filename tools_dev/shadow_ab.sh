#!/bin/bash
# same-box interleaved A/B of the persistent bf16 weight image (Adam writes it; the forward skips the arena re-cast)
out=${1:-gpurun_out/shadow_ab}
for rep in 1 2; do
  for cfg in 2 3 4; do
    for mode in off on; do
      flag="--no-persistent-shadow"; [ $mode = on ] && flag=""
      python bench.py --config $cfg --no-cpu-baseline --steps 300 --warmup 50 $flag > ${out}_cfg${cfg}_${mode}_$rep.json 2> ${out}_cfg${cfg}_${mode}_$rep.log
      echo "cfg$cfg $mode rep$rep: $(grep -h 'timed' ${out}_cfg${cfg}_${mode}_$rep.log)" | tee -a ${out}.txt
    done
  done
done
