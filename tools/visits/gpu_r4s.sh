#!/bin/bash
# visit S: XCD-contiguous tile ranges in conv_pool_h / sinc_conv0_h (traffic), profiled warm-up in bench.py
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "=== pytest -m gpu"
timeout -s KILL 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | grep -v amdgpu.ids | tail -4
for i in 1 2; do
echo "=== bench, the driver's command ($i)"
timeout -s KILL 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r4s_driver$i.json 2> gpurun_out/bench_r4s_driver$i.err
echo "exit $?"
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/bench_r4s_driver$i.json") if l.startswith('{"metric"')][-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["kernel"], "traffic", d["roofline"]["traffic"], "frac", d["roofline"]["frac"])
print("hbm", {k: v for k, v in d["hbm_gbps_step"].items() if k != "source"})
print("host", d["host"])
print("exact", d["exact_f32"]["value"], "host_fed", d["host_fed"]["value"], "rehearsal", d["host_rehearsal"]["pinned"]["value"], d["host_rehearsal"]["unpinned"]["value"])
for k in d["roofline_kernels"]:
    print("  ", k["kernel"][:44].ljust(44), k["avg_launch_us"], k["traffic"], k["alg_bytes_per_launch"])
PY
done
echo "=== bench 200 steps x2"
for i in 1 2; do
timeout -s KILL 600 python bench.py --steps 200 --warmup 10 --pmc off --no-cpu-baseline --no-rehearsal --no-exact-f32 --no-host-pass 2> /dev/null | grep '^{"metric"' | cut -c1-260
done
