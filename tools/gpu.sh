#!/bin/bash
# build everything in-tree, then run a command on the MI355X box:  tools/gpu.sh <timeout> '<cmd>'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" >/dev/null
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
