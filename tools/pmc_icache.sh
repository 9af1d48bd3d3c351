# Instruction-cache counters of the pair kernels (bench.py) and of the fused-histogram kernels (bench.py --workload rdf): requests,
# hits, misses of the shader instruction caches per launch, beside the wave cycles spent waiting for any instruction.
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for W in search_fit rdf; do
D=$R/gpurun_out/pmc_icache_$W
rm -rf $D
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $D -- python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - "$D" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "hist_kernel" in k: key = "hist_lean"
    elif "hist_plan" in k: key = "hist_plan"
    elif "pair_kernel" in k: key = "fill" if "<0, 1," in k else ("hist_rest" if "<0, 2," in k else "count")
    else: continue
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
for key in acc:
    v = {c: acc[key][c] / n[(key, c)] for c in acc[key]}
    print(key, {c: round(x / 1e6, 3) for c, x in v.items()}, "(millions per launch)",
          "miss rate %.4f" % (v.get("SQC_ICACHE_MISSES", 0) / max(v.get("SQC_ICACHE_REQ", 1), 1)),
          "wait / wave cycles %.3f" % (v.get("SQ_WAIT_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1)))
PY
done
