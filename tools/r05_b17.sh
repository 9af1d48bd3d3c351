O=gpurun_out/r05/b17
mkdir -p $O
bash tools/r05_rdf_trace.sh base molar_amd/_ab/libmolar_hip_ab.so > $O/trace_base.txt 2>&1
MOLAR_HIP_HOST_GRID_WAIT=1 bash tools/r05_rdf_trace.sh hostwait molar_amd/_ab/libmolar_hip_ab.so > $O/trace_hostwait.txt 2>&1
MOLAR_HIP_NO_SIDE_STREAM=1 bash tools/r05_rdf_trace.sh noside molar_amd/_ab/libmolar_hip_ab.so > $O/trace_noside.txt 2>&1
for f in base hostwait noside; do echo "== $f"; tail -14 $O/trace_$f.txt; done
