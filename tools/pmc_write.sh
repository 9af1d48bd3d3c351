cd /tmp && export TMPDIR=/tmp
R=/root/repo
for C in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum TCC_BUSY_sum" "TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TD_BUSY_avr"; do
  D=$R/gpurun_out/pmc_w_$(echo $C | cut -d' ' -f1)
  rm -rf $D
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $D.log 2>&1 || tail -2 $D.log
  python - "$D" <<'PY'
import csv, glob, collections, sys
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not fs: print("no output for", sys.argv[1]); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"]
    if "pair_kernel" not in k: continue
    key = "fill" if "<0, 1," in k else "count"
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
for key in acc:
    print(key, {c: round(v / n[(key, c)] / 1e6, 3) for c, v in acc[key].items()}, "(millions)")
PY
done
