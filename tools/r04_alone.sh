# per-kernel durations WITHOUT overlap: one context, one frame at a time, fit behind the search on the same stream
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r04; mkdir -p $O
rm -rf $O/alone
rocprofv3 --kernel-trace --stats --output-format csv -d $O/alone -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pipeline --serial-measure --preheat 0.5 > $O/alone_bench.json 2>/dev/null
F=$(find $O/alone -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:32]:
    print(r['Name'][:80].ljust(80), r['Calls'].rjust(6), '%10.1f us' % (float(r['AverageNs'])/1e3), r['Percentage'])
PY
tail -1 $O/alone_bench.json | cut -c1-200
rm -rf $O/alone
