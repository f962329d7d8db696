#!/bin/bash
# round 6: the overlapped optimizer (bench.py --overlap) with a bounded AdamW footprint (DXA_ADAMW_GRID) and stream priority,
# all inside ONE box.  -> gpurun_out/r06_overlap_grid.txt
out=gpurun_out/r06_overlap_grid.txt
: > $out
B="python bench.py --no-recipe --no-cpu-baseline --no-secondary --no-latency --no-dp-emulation --steps 10 --warmup 3"
run() { # label, env..., extra args after --
  label=$1; shift
  line=$(env "$@" $B $EXTRA 2>/dev/null | tail -1)
  echo "$label | $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["frac"])')" >> $out
}
EXTRA="" run "serial" A=1
EXTRA="--overlap" run "overlap full grid prio0" A=1
EXTRA="--overlap" run "overlap full grid prio low(1)" DXA_OPT_STREAM_PRIO=1
for g in 64 128 256 512 1024; do
  EXTRA="--overlap" run "overlap grid $g prio0" DXA_ADAMW_GRID=$g
done
EXTRA="--overlap" run "overlap grid 256 prio low" DXA_ADAMW_GRID=256 DXA_OPT_STREAM_PRIO=1
EXTRA="" run "serial again" A=1
cat $out
