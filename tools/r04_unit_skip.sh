# experiment: fill pass that skips the (row, chunk) units a brute-force producer found empty (library built with -DMOLAR_HIP_UNIT_SKIP)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
export MOLAR_HIP_PLUGIN=$R/molar_amd/_ab/libmolar_hip_us.so
for v in off on off on; do
  if [ $v = on ]; then export MOLAR_HIP_UNIT_SKIP_ON=1; else unset MOLAR_HIP_UNIT_SKIP_ON; fi
  rm -rf $R/gpurun_out/us_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/us_$v -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $R/gpurun_out/us_$v.json 2>/dev/null
  python - <<PY
import csv,glob,json
f=glob.glob("$R/gpurun_out/us_$v/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "pair_kernel<0, 1>" in r["Name"] or "pair_kernel<0, 0>" in r["Name"] or "rowbits" in r["Name"]:
        print("$v", r["Name"][:34], r["Calls"], round(float(r["AverageNs"])/1e6,4), "ms")
try:
    d=json.loads(open("$R/gpurun_out/us_$v.json").read().strip().splitlines()[-1]); print("$v", "verified", d["verified_against_single_context"], round(d["value"],1))
except Exception as e: print("no line", e)
PY
done
