cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/${ROUND:-r02}/fit_batch
mkdir -p $O
for SEL in stride random block; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$SEL -- python $R/tools/prof_fit_batch.py $SEL > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$SEL -- python $R/tools/prof_fit_batch.py $SEL > /dev/null 2>&1
  python - "$O" "$SEL" <<'PY'
import csv, glob, sys
o, sel = sys.argv[1:3]
dur = [float(r["AverageNs"]) for f in glob.glob(f"{o}/stats_{sel}/**/*kernel_stats.csv", recursive=True) for r in csv.DictReader(open(f)) if "k_fit_sums_packed" in r["Name"]]
fs = [float(r["Counter_Value"]) for f in glob.glob(f"{o}/fetch_{sel}/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "k_fit_sums_packed" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
if dur and fs:
    kb = sum(fs) / len(fs)            # FETCH_SIZE is in KiB-like units of 1 kB per the counter definition; x2 on gfx950 for wide reads (guide)
    print(f'{{"selection": "{sel}", "kernel": "k_fit_sums_packed<true>", "frames": 64, "avg_us": {dur[0] / 1e3:.1f}, "FETCH_SIZE_raw_kB": {kb:.0f}, "per_frame_MB_raw": {kb * 1024 / 64 / 1e6:.2f}}}')
PY
done
