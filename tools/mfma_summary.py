#!/usr/bin/env python
"""Per-kernel matrix-core utilisation from a rocprofv3 --pmc pass of SQ_VALU_MFMA_BUSY_CYCLES,
SQ_BUSY_CYCLES and GRBM_GUI_ACTIVE.  ROCm 7.2 ships no gfx950 derived-counter section
(MI355X_MICROARCH.md, PMC slots), so the gfx94x MfmaUtil formula is applied by hand:
    util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)
SQ_VALU_MFMA_BUSY_CYCLES sums SIMD-busy cycles over the chip (32 per v_mfma_f32_16x16x4_f32: the
TDNN launches show 3.8e8 = #MFMA x 32), GRBM_GUI_ACTIVE is reported summed over the 8 XCDs
(5.2e6 "cycles" for a 0.3 ms kernel).  Cross-check: TDNN util 0.57 by this formula vs 0.56 from
algorithmic FLOP / duration / 157.3 TF.  Raw sums are kept next to it.  usage: mfma_summary.py <pmc_dir> <out.json>"""
import csv, glob, json, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[r["Kernel_Name"]] += 1
out = {}
for k, c in acc.items():
    gui, mf = c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    out[k] = {"launches": n[k], "mfma_busy_cycles": mf, "sq_busy_cycles": c.get("SQ_BUSY_CYCLES", 0.0),
              "gui_active_cycles": gui, "mfma_util": round(mf / (gui / 8 * 256 * 4), 4) if gui else None}
json.dump({"note": "sums over all launches of `bench.py --steps 3`; util = MFMA_BUSY / (GUI_ACTIVE / 8 * 256 * 4)",
           "kernels": out}, open(sys.argv[2], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles"])[:12]:
    print(f"{v['mfma_util']}  x{v['launches']:3d}  {k[:90]}")
