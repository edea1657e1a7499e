#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
for d in 0 1 8 9 4 2 6 13 15; do
  echo -n "DZ_CONV0_DBG=$d  "; DZ_CONV0_DBG=$d timeout 120 python tools/kbench.py --only sinc_conv0_split --reps 30 2>&1 | grep "sinc_conv0_split" | head -1
done
