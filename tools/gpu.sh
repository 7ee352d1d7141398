#!/bin/bash
# Stamp the tree's commit into .ams_commit (travels with the snapshot; the GPU box has no .git) and run a command on the GPU box,
# retrying while the pod's slots are busy:   tools/gpu.sh <timeout_s> '<command>'
cd /root/repo
c=$(git rev-parse --short HEAD)
git diff --quiet HEAD -- . ':!profiles' ':!*.md' || c="$c-dirty"
echo $c > .ams_commit
exec tools/gpurun_retry.sh "$@"
