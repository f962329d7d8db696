#!/bin/bash
# full GPU verification of the tree: every GPU test, the smoke entry, the driver-style bench line
set -x
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; tail -c 2500 gpurun_out/r03_bench.json
