cd /tmp && export TMPDIR=/tmp
R=/root/repo
mkdir -p $R/gpurun_out/r01b
python $R/bench.py > $R/gpurun_out/r01b/bench.json 2> $R/gpurun_out/r01b/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01b/stats -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01b/fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01b/write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
tail -1 $R/gpurun_out/r01b/bench.json | cut -c1-600
find $R/gpurun_out/r01b -name "*.csv" | head -20
