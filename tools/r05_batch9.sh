O=gpurun_out/r05/b9
mkdir -p $O
timeout 600 python tools/fuzz_search.py 300 5301 2>&1 | tail -1 > $O/fuzz.txt
bash tools/ab_rdf.sh molar_amd/_ab/libmolar_hip_v4.so molar_amd/libmolar_hip.so molar_amd/_ab/libmolar_hip_w8.so molar_amd/_ab/libmolar_hip_cu24w8.so molar_amd/_ab/libmolar_hip_cu24w12.so molar_amd/_ab/libmolar_hip_cu16w8.so > $O/ab.txt 2>&1
cat $O/fuzz.txt $O/ab.txt
