O=gpurun_out/r05/b20
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/$O/vdw_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/vdw_prof -- python $R/tools/bench_vdw.py > /dev/null 2>&1
F=$(find $R/$O/vdw_prof -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]: print(r['Name'][:90].ljust(90), r['Calls'], '%.1f us avg' % (float(r['AverageNs'])/1e3), r['Percentage'])
PY
rm -rf $R/$O/vdw_prof
