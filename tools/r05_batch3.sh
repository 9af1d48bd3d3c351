set -x
O=gpurun_out/r05/b3
mkdir -p $O
for lib in dbg dbglds; do
MOLAR_HIP_DEBUG_LAUNCH=8 MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_$lib.so python tools/hist_wave_times.py > $O/waves_$lib.txt 2>&1
MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_$lib.so python tools/hist_wave_times.py > $O/waves_last_$lib.txt 2>&1
done
bash tools/ab_rdf.sh molar_amd/_ab/libmolar_hip_r04.so molar_amd/_ab/libmolar_hip_dbg.so molar_amd/_ab/libmolar_hip_dbglds.so > $O/ab.txt 2>&1
tail -n 3 $O/*.txt
