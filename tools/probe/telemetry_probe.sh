# What this box offers for clocks / power / partition-mode telemetry (round 4, VERDICT r03 #1b)
mkdir -p gpurun_out
{
echo "== sysfs"; for d in /sys/class/drm/card*/device; do echo $d; ls $d | tr '\n' ' '; echo; for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk current_compute_partition current_memory_partition available_compute_partition available_memory_partition power_dpm_force_performance_level mem_info_vram_total mem_info_vram_used unique_id vbios_version; do [ -r $d/$f ] && { echo "-- $f"; cat $d/$f; }; done; ls $d/hwmon/*/ 2>/dev/null | tr '\n' ' '; echo; for f in $d/hwmon/*/power1_average $d/hwmon/*/power1_input $d/hwmon/*/power1_cap $d/hwmon/*/power1_cap_max $d/hwmon/*/temp*_input $d/hwmon/*/freq*_input; do [ -r $f ] && echo "$f $(cat $f)"; done; done
echo "== which"; which amd-smi rocm-smi rocminfo
echo "== amd-smi"; timeout 30 amd-smi metric --json 2>&1 | head -150
echo "== amd-smi static partition"; timeout 30 amd-smi static --partition --json 2>&1 | head -60
echo "== rocm-smi"; timeout 30 rocm-smi --showclocks --showpower --showmaxpower --showmemorypartition --showcomputepartition --showperflevel --json 2>&1 | head -60
echo "== python amdsmi"; python -c "import amdsmi; print(amdsmi.__file__)" 2>&1
} > gpurun_out/r04_telemetry_probe.log 2>&1
tail -5 gpurun_out/r04_telemetry_probe.log
