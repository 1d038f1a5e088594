#!/bin/bash
# gpurun wrapper that tells the GPU box which commit it runs (the snapshot has no .git): writes the short hash -- with
# "+dirty" when the work tree differs from HEAD -- and a fingerprint of the source files to .head_commit; bench.py puts
# the hash into its JSON line only when the fingerprint matches the files it runs from (a stale file prints nothing).
#   tools/gpu.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.."
# the last commit that touches the product path (hypelcnn_amd/, include/, bench.py): what bench.py itself reports with a .git
h=$(git log -1 --format=%h -- hypelcnn_amd include bench.py)
git diff --quiet HEAD -- hypelcnn_amd include bench.py || h="$h+dirty"
echo "$h $(python -c 'import bench; print(bench.source_fingerprint())')" > .head_commit
exec /usr/local/graft/bin/gpurun "$@"
