#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for cfg in "0 8 1" "0 8 0" "256 8 1" "256 8 0" "0 8 1" "0 8 0"; do set -- $cfg; echo "== target $1 minkb $2 pair $3"; if [ $3 = 0 ]; then export DXA_SKINNY_NOPAIR=1; else unset DXA_SKINNY_NOPAIR; fi; DXA_SKINNY_TARGET=$1 DXA_SKINNY_MINKB=$2 python scripts/skinny_bench.py 2>&1 | grep "us/call\|rror"; done
