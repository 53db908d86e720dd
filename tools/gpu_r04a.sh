#!/bin/bash
# round 4, GPU session a: the strict trunk's first run — denormal probe, parity tests of the net, probe timings,
# telemetry discovery for the bench (what clock / power sources the box offers)
O=gpurun_out/r04a; mkdir -p $O
./tools/bin/mfma_denorm_probe > $O/denorm.txt 2>&1; cat $O/denorm.txt
timeout 900 python -m pytest tests/test_net.py -m gpu -x -q -s 2>&1 | tail -60 > $O/test_net.txt; tail -40 $O/test_net.txt
timeout 600 python tools/strict_probe.py 8192 10 > $O/strict_probe.txt 2>&1; cat $O/strict_probe.txt
( ls -la /sys/class/drm/ ; for h in /sys/class/drm/card*/device/hwmon/hwmon*; do echo "== $h"; ls $h; for f in $h/power1_average $h/power1_input $h/freq1_input $h/freq1_label $h/power1_label; do [ -e $f ] && echo "$f: $(cat $f)"; done; done
  for c in /sys/class/drm/card*/device; do echo "== $c"; cat $c/pp_dpm_sclk 2>/dev/null | head -5; done
  timeout 20 rocm-smi --showpower --showclocks --json 2>&1 | head -30
  timeout 20 amd-smi metric --power --clock --json 2>&1 | head -60
  python -c "import amdsmi; print('amdsmi importable', amdsmi.__file__)" 2>&1 | tail -1 ) > $O/telemetry.txt 2>&1
head -80 $O/telemetry.txt
timeout 600 python bench.py --dtype strict --steps 400 --steady-steps 600 --no-cpu-baseline > $O/bench_strict.json 2> $O/bench_strict.err; python tools/jline.py $O/bench_strict.json | head -40
