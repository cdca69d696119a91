#!/bin/bash
# usage: tools/grun.sh <timeout seconds> '<command>'
# Runs <command> on the GPU box inside a FROZEN copy of the tree as of the moment this script starts: gpurun takes its snapshot
# only after queueing for a slot, so edits made while a call waits would otherwise travel half-built.  The tree is packed into
# .frozen/<id>.tar (git-ignored, pushed with the snapshot), unpacked under /tmp/w on the box, and gpurun_out/ there is a link to
# the real output directory that gpurun merges back.  Retries while the pool is busy (exit 3: nothing charged).
t=$1; shift
cmd="$*"
cd "$(dirname "$0")/.." || exit 1
id=$(date +%s)_$$
mkdir -p .frozen && rm -f .frozen/*.tar
tar cf .frozen/$id.tar --exclude=./.git --exclude=./gpurun_out --exclude=./.frozen --exclude=__pycache__ --exclude='*.o' --exclude=./.pytest_cache .
for try in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "export OUT=\$GRAFT_REPO_ROOT/gpurun_out && mkdir -p \$OUT/r3 /tmp/w && tar xf .frozen/$id.tar -C /tmp/w && cd /tmp/w && ln -s \$OUT gpurun_out && export GRAFT_REPO_ROOT=/tmp/w && $cmd"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 75
done
exit 3
