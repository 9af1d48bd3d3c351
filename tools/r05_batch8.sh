O=gpurun_out/r05/b8
mkdir -p $O
MOLAR_HIP_DEBUG_LAUNCH=8 MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_dbgsync.so python tools/hist_wave_times.py 2>/dev/null > $O/waves.txt
cat $O/waves.txt
