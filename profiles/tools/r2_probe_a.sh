set -x
mkdir -p gpurun_out/r2a
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2a/smoke.log 2>&1
python profiles/tools/mfma_rate_probe.py > gpurun_out/r2a/rate.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench_s20.json 2> gpurun_out/r2a/bench_s20.err
python bench.py --cpu-seconds 0 > gpurun_out/r2a/bench_s1000.json 2> gpurun_out/r2a/bench_s1000.err
(ls -l /sys/class/drm/; ls /sys/class/drm/card*/device/hwmon/*/; cat /sys/class/drm/card*/device/hwmon/*/freq1_input; cat /sys/class/drm/card*/device/hwmon/*/power1_average; cat /sys/class/drm/card*/device/pp_dpm_sclk; rocm-smi --showclocks --showpower; amd-smi metric -c -p 2>&1 | head -60) > gpurun_out/r2a/sysfs.log 2>&1
tail -3 gpurun_out/r2a/smoke.log; cat gpurun_out/r2a/rate.log; cat gpurun_out/r2a/bench_s20.json; cat gpurun_out/r2a/bench_s1000.json
