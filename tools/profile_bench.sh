cd /tmp && export TMPDIR=/tmp
R=/root/repo
T=${ROUND:-r02}          # round tag: raw output under gpurun_out/$T, tools/refresh_profiles.py copies the summaries into profiles/
mkdir -p $R/gpurun_out/$T
python $R/bench.py > $R/gpurun_out/$T/bench.json 2> $R/gpurun_out/$T/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/stats -- python $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$T/fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$T/write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
# C4-shaped workload (fused histogram): bench line + kernel stats
python $R/bench.py --workload rdf --steps 1024 --warmup 64 --no-cpu-baseline > $R/gpurun_out/$T/rdf_bench.json 2> $R/gpurun_out/$T/rdf_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/rdf_stats -- python $R/bench.py --workload rdf --steps 256 --warmup 32 --no-cpu-baseline > /dev/null 2>&1
tail -1 $R/gpurun_out/$T/bench.json | cut -c1-600
find $R/gpurun_out/$T -name "*.csv" | head -20
