#!/bin/bash
# driver-style configs[1] line (value, ms per step, kernel ms) a few times per setting of one environment knob, alternating:
# tools/ab_c1.sh KNOB valueA valueB [repeats]
knob=$1; a=$2; b=$3; n=${4:-3}
for i in $(seq 1 $n); do
  for v in $a $b; do
    env $knob=$v python bench.py --steps 20 --warmup 5 --c3-steps 0 --sw-steps 0 --smem-steps 0 --bwasw-steps 0 --e2e-regions -1 --no-cpu-baseline |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$knob=$v', round(d['value']), round(d['ms_per_step']*1e3,1), round(d['roofline']['kernel_ms']*1e3,1), round((d['ms_per_step']-d['roofline']['kernel_ms'])*1e3,1))"
  done
done
