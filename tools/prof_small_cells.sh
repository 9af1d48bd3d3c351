# Kernel times and SQ counters of the small-cell kernels (pair_small.hip) on the 1M-atom frame at a contact cutoff:
#   bash tools/prof_small_cells.sh [rc = 0.35]   -> gpurun_out/r05/small_cells.txt   (profiles/r05_small_cells.txt)
RC=${1:-0.35}
R=/root/repo
mkdir -p $R/gpurun_out/r05
OUT=$R/gpurun_out/r05/small_cells.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sc_stats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sc_stats -- python $R/tools/bench_cutoff_sweep.py $RC > /dev/null 2>&1
python - $(find /tmp/sc_stats -name "*kernel_stats.csv" | head -1) > $OUT <<'PY'
import csv, sys
print("kernel                                                         calls    avg us    share")
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f'{r["Name"].replace("(anonymous namespace)::", "").replace("mh::pairk::", "")[:60]:60s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:9.1f}  {float(r["Percentage"]):6.2f}%')
PY
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  rm -rf /tmp/sc_pmc; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/sc_pmc -- python $R/tools/bench_cutoff_sweep.py $RC > /dev/null 2>&1
  python - $(find /tmp/sc_pmc -name "*counter_collection.csv" | head -1) >> $OUT <<'PY'
import sys, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "small_pair" not in k: continue
    key = "fill " if "<true" in k else "count"
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
for key in sorted(acc): print(key, {c: f"{v / n[(key, c)]:.4g}" for c, v in sorted(acc[key].items())})
PY
done
find $R/gpurun_out -name "*kernel_trace.csv" -delete
cat $OUT
