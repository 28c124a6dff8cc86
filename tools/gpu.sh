#!/bin/bash
# Rebuild the library (and the checker), then hand the command to gpurun: a stale .so re-tests the OLD code (NOTES.md).
# Usage: bash tools/gpu.sh <timeout-seconds> '<command>'
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
python -c "import __graft_entry__ as g; g.build()"
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
