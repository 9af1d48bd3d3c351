# kernel + memory-copy statistics of the C5 frame chain on ONE engine context (bench.py --workload membrane --streams 1): which launches a
# frame is made of, how long each takes.  usage: tools/r06_membrane_trace.sh TAG   -> gpurun_out/r06/TAG_membrane_{kernels,copies}.csv, TAG_membrane.json
R=/root/repo; T=${1:-m}; O=$R/gpurun_out/r06; mkdir -p $O; D=/tmp/mtrace_$T; rm -rf $D
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $D -- python $R/bench.py --workload membrane --streams 1 --steps 128 --warmup 16 > $O/${T}_membrane.json 2> $O/${T}_membrane.err
F=$(find $D -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cut -c1-160 "$F" | head -60 > $O/${T}_membrane_kernels.csv
F=$(find $D -name "*memory_copy_stats.csv" | head -1)
[ -n "$F" ] && head -20 "$F" > $O/${T}_membrane_copies.csv
rm -rf $D
cat $O/${T}_membrane_kernels.csv $O/${T}_membrane_copies.csv 2>/dev/null </dev/null | head -70
python -c "import json,sys;l=json.loads(open('$O/${T}_membrane.json').read().strip().splitlines()[-1]);print(l['value'],l['ms_per_step'])" </dev/null
