#!/bin/bash
# job-granularity sweep for the PairHMM C1 bench (run on the GPU box)
for tj in 2048 3072 4096 6144 8192 12288 16384 32768; do
  for ms in 256; do
    echo -n "target_jobs=$tj min_steps=$ms: "
    ACCG_PHMM_TARGET_JOBS=$tj ACCG_PHMM_MIN_STEPS=$ms python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['jobs'], 'jobs', round(d['roofline']['kernel_ms'],4),'ms kernel', round(d['value']),'GCUPS')"
  done
done
