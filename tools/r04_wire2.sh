#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for th in 0 8; do python3 tools/r04_wire2.py $th 2>&1 | tail -1; done
bash tools/r04_e2e.sh
