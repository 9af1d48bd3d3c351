# whole GPU suite + the driver's bench line + a 60-step line (gpurun_out/r04/<tag>_*)
R=/root/repo; T=${1:-chk}; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/${T}_tests.txt; cat $O/${T}_tests.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench20.json 2> $O/${T}_bench20.err
python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $O/${T}_bench60.json 2>> $O/${T}_bench20.err
for f in $O/${T}_bench20.json $O/${T}_bench60.json; do tail -1 $f | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']
print(round(d['value'],1), 'ms/step %.4f' % d['ms_per_step'], {a: round(b,4) for a,b in k.items()}, 'verified', d.get('verified_against_single_context'))"; done
tail -3 $O/${T}_bench20.err
