set -x
O=gpurun_out/r05/b5
mkdir -p $O
MOLAR_HIP_DEBUG_LAUNCH=8 MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_dbg.so python tools/hist_wave_times.py > $O/waves_dbg.txt 2>&1
MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_dbg.so python tools/hist_wave_times.py > $O/waves_last_dbg.txt 2>&1
cat $O/*.txt
