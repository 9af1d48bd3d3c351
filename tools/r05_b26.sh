O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/b26_tests.txt
timeout 600 python tools/fuzz_search.py 800 5701 2>&1 | tail -1 >> $O/b26_tests.txt
bash tools/ab_rdf.sh molar_amd/_ab/libmolar_hip_v6.so molar_amd/libmolar_hip.so > $O/b26_ab_rdf.txt 2>&1
REPS=2 bash tools/ab_bench.sh molar_amd/_ab/libmolar_hip_v6.so molar_amd/libmolar_hip.so > $O/b26_ab.txt 2>&1
cat $O/b26_tests.txt $O/b26_ab_rdf.txt $O/b26_ab.txt
