#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
TAG=${1:-r4c}
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gemm_pre" -p no:cacheprovider 2>&1 | tail -3
timeout 400 python tools/g2bench.py --out gpurun_out/g2bench_$TAG.json ${G2B_ARGS:---only tdnn2,tdnn4,proj} 2>&1 | grep -v amdgpu.ids | cut -c1-80 | tail -4
python - <<PY
import json
d=json.load(open('gpurun_out/g2bench_$TAG.json'))
for k,v in d.items():
    if not isinstance(v,dict): print(k,v); continue
    print(k, v['shape'], 'ideal', v['ideal_us_f16x3'], 'err g1 %.2e g2 %.2e g3_mt4 %.2e' % (v['g1_rel_l2_vs_f64'], v['g2_mt2_rel_l2_vs_f64'], v['g3_mt4_rel_l2_vs_f64']))
    for tag in ('g1','g2_mt2','g2_mt3','g2_mt4','g3_mt2','g3_mt3','g3_mt4'):
        print('   %-7s alone %6.1f  rec1 %6.1f  rec2 %6.1f' % (tag, v[f'{tag}_us_rec0'], v.get(f'{tag}_us_rec1',0), v.get(f'{tag}_us_rec2',0)))
PY
i=0
for cfg in ${GRID:-DZ_GEMM_GEN=1 DZ_GEMM_GEN=2 DZ_GEMM_GEN=1 DZ_GEMM_GEN=2}; do
  i=$((i+1))
  env $(echo $cfg | tr ',' ' ') timeout 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-exact-f32 --pmc off --no-host-pass \
      > gpurun_out/bench_${TAG}_$i.json 2>gpurun_out/bench_${TAG}_$i.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/bench_${TAG}_$i.json"))
    print("$cfg value", d["value"], "ms/step", d["ms_per_step"], " | ".join("%s %.0f" % (k["kernel"][:22], k["avg_launch_us"]) for k in d["roofline_kernels"][:9]))
except Exception as e:
    print("bench failed:", e)
PY
done
