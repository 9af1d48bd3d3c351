O=gpurun_out/r05/b24
mkdir -p $O
bash tools/ab_rdf.sh molar_amd/libmolar_hip.so molar_amd/_ab/libmolar_hip_ta512.so molar_amd/_ab/libmolar_hip_ta2k.so > $O/ab.txt 2>&1
MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_ta512.so timeout 300 python tools/fuzz_search.py 200 5601 2>&1 | tail -1 >> $O/ab.txt
cat $O/ab.txt
