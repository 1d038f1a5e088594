#!/bin/bash
# gpurun wrapper that tells the GPU box which commit it runs (the snapshot has no .git): writes the short hash -- with
# "+dirty" when the work tree differs from HEAD -- and a fingerprint of the source files to .head_commit; bench.py puts
# the hash into its JSON line only when the fingerprint matches the files it runs from (a stale file prints nothing).
#   tools/gpu.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.."
h=$(git rev-parse --short HEAD)
git diff --quiet HEAD -- . ':!.head_commit' || h="$h+dirty"
echo "$h $(python -c 'import bench; print(bench.source_fingerprint())')" > .head_commit
exec /usr/local/graft/bin/gpurun "$@"
