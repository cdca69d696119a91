#!/bin/bash
# stream-budget sweep on the configs[3]-like batch: tools/sweep_c3.sh [lib]
for b in 300 400 512 640 800 1024 1300 1700 2048 4096; do
  echo -n "budget $b: "; ACCG_PHMM_STREAM_BUDGET=$b python tools/bench_c3.py 1024 2>&1 | grep -E "^fast|jobs" | tr '\n' ' '; echo
done
