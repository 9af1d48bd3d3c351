cd /tmp && export TMPDIR=/tmp
R=/root/repo
D=$R/gpurun_out/pmc_valu
rm -rf $D
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $D -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/root/repo/gpurun_out/pmc_valu/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "pair_kernel" not in k: continue
    key = "fill" if "<0, 1," in k else "count"
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
for key in acc:
    print(key, {c: round(v / n[(key, c)] / 1e6, 2) for c, v in acc[key].items()}, "(millions)")
PY
