set -x
O=gpurun_out/r05/b1
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "hist or c4_shape" 2>&1 | tail -5 > $O/tests.txt
timeout 600 python tools/fuzz_search.py 500 5001 2>&1 | tail -3 > $O/fuzz.txt
bash tools/ab_rdf.sh molar_amd/_ab/libmolar_hip_r04.so molar_amd/libmolar_hip.so > $O/ab.txt 2>&1
MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_dbg.so SKIPS="0 1 2 4 3" bash tools/dbg_skip_rdf.sh > $O/skip.txt 2>&1
MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_dbg.so python tools/hist_wave_times.py > $O/waves.txt 2>&1
cat $O/tests.txt $O/fuzz.txt $O/ab.txt $O/skip.txt $O/waves.txt
