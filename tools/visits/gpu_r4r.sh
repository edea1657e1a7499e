#!/bin/bash
# visit R: (1) the driver's own command after the symbol fix (traffic of the headline kernel must not be null);
# (2) TCC counters of the isolated SincNet stage-1/2 kernels (review item 7: explain conv_pool_h<80>'s 1.49x)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "=== bench, the driver's command"
timeout -s KILL 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r4r_driver.json 2> gpurun_out/bench_r4r_driver.err
echo "exit $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_r4r_driver.json") if l.startswith('{"metric"')][-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["kernel"], "traffic", d["roofline"]["traffic"], "frac", d["roofline"]["frac"])
print("hbm", {k: v for k, v in d["hbm_gbps_step"].items() if k != "source"})
PY
echo "=== TCC probe"
bash tools/tcc_probe.sh r4r wave_stats,conv1_pool,conv2_pool 2>&1 | cut -c1-600
