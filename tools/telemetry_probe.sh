#!/bin/bash
# what telemetry can an ordinary user read on the GPU box?
O=gpurun_out/r05a_telemetry.txt
{
echo "== id"; id
echo "== drm cards"; ls -d /sys/class/drm/card*/device 2>&1
for d in /sys/class/drm/card*/device; do
  echo "== $d"; ls $d 2>&1 | tr '\n' ' '; echo
  for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk power_dpm_force_performance_level gpu_busy_percent mem_busy_percent pp_power_profile_mode; do
    echo "-- $f"; cat $d/$f 2>&1 | head -20
  done
  for h in $d/hwmon/hwmon*; do
    echo "== $h"; ls $h | tr '\n' ' '; echo
    for f in $h/power1_average $h/power1_input $h/power1_cap $h/freq1_input $h/freq2_input $h/temp1_input $h/temp2_input $h/temp3_input $h/energy1_input; do echo "-- $f"; cat $f 2>&1; done
  done
  echo "-- gpu_metrics size"; wc -c $d/gpu_metrics 2>&1
done
echo "== rocm-smi"; timeout 60 rocm-smi --showpower --showclocks --showperflevel --showtemp 2>&1 | head -60
echo "== amd-smi metric"; timeout 60 amd-smi metric -g 0 --power --clock --temperature 2>&1 | head -80
echo "== amd-smi static"; timeout 60 amd-smi static -g 0 --limit 2>&1 | head -40
echo "== python amdsmi"; python3 -c "import amdsmi; print(amdsmi.__file__)" 2>&1
} > $O 2>&1
echo done
