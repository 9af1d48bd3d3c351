# SQ counters of the fused-histogram kernels (bench.py --workload rdf, frames form: every hist_kernel launch carries 16 frames -
# the figures are per LAUNCH, divide by 16 for a frame), one rocprofv3 pass per counter group
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN"; do
  D=$R/gpurun_out/pmch_$(echo $C | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python $R/bench.py --workload rdf --steps 32 --warmup 16 --profile-steps 16 --no-cpu-baseline > /dev/null 2>&1
  F=$(find $D -name "*counter_collection.csv" | head -1)
  python - "$F" <<'PY'
import sys, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no counters:", e); rows = []
for r in rows:
    k = r["Kernel_Name"]
    if "hist_kernel" in k: key = "hist_lean"
    elif "pair_kernel" in k: key = "hist_rest"
    else: continue
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(key, r["Counter_Name"])] += 1
for key in acc:
    print(key, {c: v / n[(key, c)] for c, v in acc[key].items()})
PY
done
