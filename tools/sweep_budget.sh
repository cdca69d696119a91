#!/bin/bash
# usage (GPU box): tools/sweep_budget.sh   -- configs[1] kernel time against the haplotype-stream budget per job
for bud in 303 604 905 1206 1507 2410 4000; do
  echo -n "budget=$bud: "
  ACCG_PHMM_STREAM_BUDGET=$bud python bench.py --steps 100 --warmup 10 --no-cpu-baseline --sw-steps 0 --smem-steps 0 --bwasw-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['jobs'], 'jobs', round(d['roofline']['kernel_ms'],4),'ms kernel', round(d['value']),'GCUPS')"
done
