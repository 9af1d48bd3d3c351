cd /tmp && export TMPDIR=/tmp
R=/root/repo
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVES"; do
  D=$R/gpurun_out/pmc_$(echo $C | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  F=$(find $D -name "*counter_collection.csv" | head -1)
  python - "$F" <<'PY'
import sys, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "pair_kernel" not in k: continue
    key = "fill" if "<0, 1," in k else "count"
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); 
    n[(key, r["Counter_Name"])] += 1
for key in acc:
    print(key, {c: v / n[(key, c)] for c, v in acc[key].items()})
PY
done
