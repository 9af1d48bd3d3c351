set -x
O=gpurun_out/r05/b4
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "hist or c4_shape" 2>&1 | tail -5 > $O/tests.txt
timeout 600 python tools/fuzz_search.py 600 5101 2>&1 | tail -3 > $O/fuzz.txt
MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_w8.so timeout 600 python tools/fuzz_search.py 300 5102 2>&1 | tail -3 > $O/fuzz_w8.txt
bash tools/ab_rdf.sh molar_amd/libmolar_hip.so molar_amd/_ab/libmolar_hip_w8.so > $O/ab.txt 2>&1
for lib in dbg dbgw8; do
MOLAR_HIP_DEBUG_LAUNCH=8 MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_$lib.so python tools/hist_wave_times.py > $O/waves_$lib.txt 2>&1
MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_$lib.so python tools/hist_wave_times.py > $O/waves_last_$lib.txt 2>&1
done
MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_dbg.so SKIPS="0 1 2 4 3 7" bash tools/dbg_skip_rdf.sh > $O/skip.txt 2>&1
bash tools/r05_rdf_trace.sh new > $O/trace_new.txt 2>&1
tail -n 4 $O/tests.txt $O/fuzz.txt $O/fuzz_w8.txt $O/ab.txt $O/skip.txt
