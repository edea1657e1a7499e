#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
for cfg in DZ_WAIT=spin DZ_WAIT=block DZ_WAIT=spin DZ_WAIT=block; do
  env $cfg timeout 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-exact-f32 --pmc off --no-host-pass --no-rehearsal 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'], d['host'])"
done
echo "=== driver form with rehearsal"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc all --no-cpu-baseline > gpurun_out/bench_r4m_driver.json 2> gpurun_out/bench_r4m_driver.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r4m_driver.json"))
print("value", d["value"], "ms/step", d["ms_per_step"])
print("mfma_util_step", {k: v for k, v in d["mfma_util_step"].items() if k not in ("note", "source")})
print("exact mfma_util_step", {k: v for k, v in d["exact_f32"]["mfma_util_step"].items() if k not in ("note", "source")})
print("hbm_gbps_step", {k: v for k, v in d["hbm_gbps_step"].items() if k != "source"})
print("host", {k: v for k, v in d["host"].items() if k != "note"})
r = d["host_rehearsal"]
print("rehearsal", r["cores_per_rank"], r["pinned_over_unpinned"], r["pinned"], r["unpinned"])
PY
