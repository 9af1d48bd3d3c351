O=gpurun_out/r05/b19
mkdir -p $O
REPS=2 bash tools/ab_bench.sh molar_amd/libmolar_hip.so molar_amd/_ab/libmolar_hip_cs2.so molar_amd/_ab/libmolar_hip_cs4.so > $O/ab.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/tests.txt
cat $O/ab.txt $O/tests.txt
