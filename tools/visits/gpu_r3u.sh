#!/bin/bash
# same-visit A/B of two checkouts: the working tree vs _ab_old (git worktree of an earlier commit, built in place)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
for i in 1 2 3 4; do
  for side in new old; do
    b=bench.py; [ $side = old ] && b=_ab_old/bench.py
    timeout 90 python $b --steps 200 --warmup 10 --no-cpu-baseline --no-exact-f32 --no-host-pass --pmc off \
        > gpurun_out/ab_${side}_$i.json 2> gpurun_out/ab_${side}_$i.err
    python - <<PY
import json
d = json.load(open("gpurun_out/ab_${side}_$i.json"))
ks = {k["kernel"][:28]: k["avg_launch_us"] for k in d["roofline_kernels"][:9]}
print("$side $i", d["ms_per_step"], " ".join("%s=%.0f" % (k.split("_kernel")[0][-14:] + k.split("_kernel")[1][:6], v) for k, v in ks.items()))
PY
  done
done
