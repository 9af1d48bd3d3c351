# round-4 starting point on one box: the driver's bench line, a kernel trace for tools/timeline.py, and the SQ counters of
# the fused-histogram kernels (profiles/r04_hist_sq_mix.txt)
R=/root/repo
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench20_start.json 2> $O/bench20_start.err
tail -1 $O/bench20_start.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace_start -- python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
F=$(find $O/trace_start -name "*kernel_trace.csv" | head -1)
python $R/tools/timeline.py $F > $O/timeline_start.txt 2>&1
cat $O/timeline_start.txt
rm -rf $O/trace_start
bash $R/tools/pmc_hist_kernel.sh > $O/hist_sq_mix.txt 2>&1
cat $O/hist_sq_mix.txt
