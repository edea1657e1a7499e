# GPU clock / power while the 64-stream pipeline runs (is the in-pipeline slowdown of the recurrence the clock?)
mkdir -p gpurun_out/${CLK_TAG:-r06zl}
python - > gpurun_out/${CLK_TAG:-r06zl}/sysfs.txt 2>&1 <<'PY' &
import glob, time, os
files = []
for pat in ("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_average",
            "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input", "/sys/class/drm/card*/device/gpu_busy_percent"):
    files += sorted(glob.glob(pat))
print("files", files, flush=True)
t_end = time.time() + 150
while time.time() < t_end and not os.path.exists("gpurun_out/${CLK_TAG:-r06zl}/stop"):
    vals = []
    for f in files:
        try:
            vals.append(open(f).read().strip())
        except Exception as e:
            vals.append("x")
    print(f"{time.time():.3f}", *vals, flush=True)
    time.sleep(0.05)
PY
SMI=$!
rm -f gpurun_out/${CLK_TAG:-r06zl}/stop
date +%s.%N > gpurun_out/${CLK_TAG:-r06zl}/t_start.txt
python bench.py --steps ${CLK_STEPS:-2400} --warmup 10 ${CLK_ARGS:-} --pmc off --no-cpu-baseline --no-rehearsal --no-exact-f32 --no-host-pass --details gpurun_out/${CLK_TAG:-r06zl}/d.json > gpurun_out/${CLK_TAG:-r06zl}/b.json 2> gpurun_out/${CLK_TAG:-r06zl}/b.err
date +%s.%N > gpurun_out/${CLK_TAG:-r06zl}/t_end.txt
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -4
touch gpurun_out/${CLK_TAG:-r06zl}/stop
wait $SMI
grep -E "timed region|streams resident|warm-up done|^\[bench \+ +0.0" gpurun_out/${CLK_TAG:-r06zl}/b.err | cut -c1-200
cut -c1-120 gpurun_out/${CLK_TAG:-r06zl}/b.json
wc -l gpurun_out/${CLK_TAG:-r06zl}/sysfs.txt; head -2 gpurun_out/${CLK_TAG:-r06zl}/sysfs.txt | cut -c1-400
