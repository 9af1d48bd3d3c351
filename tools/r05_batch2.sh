set -x
O=gpurun_out/r05/b2
mkdir -p $O
bash tools/r05_rdf_trace.sh new > $O/trace_new.txt 2>&1
bash tools/r05_rdf_trace.sh r04 molar_amd/_ab/libmolar_hip_r04.so > $O/trace_r04.txt 2>&1
for v in 0 1; do
if [ $v = 1 ]; then export MOLAR_HIP_NO_SIDE_STREAM=1; else unset MOLAR_HIP_NO_SIDE_STREAM; fi
MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_dbg.so python bench.py --workload rdf --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']; print('noside=$v', round(d['value'],1), 'grid %.3f hist %.3f' % (k['grid_build'], k['pair_fill']))" >> $O/noside.txt
done
cat $O/noside.txt
