#!/bin/bash
# gpurun wrapper that tells the GPU box which commit it runs (the snapshot has no .git): writes the short hash -- with
# "+dirty" when the work tree differs from HEAD -- to .head_commit, which bench.py puts into its JSON line.
#   tools/gpu.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.."
h=$(git rev-parse --short HEAD)
git diff --quiet HEAD -- . ':!.head_commit' || h="$h+dirty"
echo "$h" > .head_commit
exec /usr/local/graft/bin/gpurun "$@"
