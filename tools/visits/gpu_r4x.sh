#!/bin/bash
# visit X: whole GPU suite + the driver's command on the code with k_gemm_f32.hip
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout -s KILL 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | grep -v amdgpu.ids | tail -4
timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -s KILL 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc all > gpurun_out/bench_r4x_driver.json 2> gpurun_out/bench_r4x_driver.err
echo "exit $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_r4x_driver.json") if l.startswith('{"metric"')][-1])
print("value", d["value"], "ms", d["ms_per_step"], "| exact", d["exact_f32"]["value"], d["exact_f32"]["ms_per_step"], "| host_fed", d["host_fed"]["value"])
e = d["exact_f32"]
print("exact roofline", e["roofline"]["kernel"], e["roofline"]["frac"], "mfma", e["roofline_mfma"]["kernel"], e["roofline_mfma"]["frac"], e["roofline_mfma"]["avg_launch_us"], e["roofline_mfma"].get("traffic"))
print("exact step", e["mfma_util_step"]["busy_frac_pmc"], e["mfma_util_step"]["frac_of_peak"], e["hbm_gbps_step"]["bytes_per_step"])
PY
