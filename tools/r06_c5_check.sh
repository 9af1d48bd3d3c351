# C5 after a change to the membrane chain: its GPU tests, the three fuzzers that cover it, and the leg on one and on four contexts.
# usage: tools/r06_c5_check.sh TAG  -> gpurun_out/r06/TAG_c5.txt
R=/root/repo; T=${1:-c5}; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
{
timeout 900 python -m pytest tests/test_gpu_membrane.py tests/test_gpu_membrane_frame.py tests/test_gpu_measure.py tests/test_gpu_membrane_fixture.py -x -q 2>&1 | tail -3
timeout 300 python tools/fuzz_lipid_order.py 300 61 2>&1 | tail -2
timeout 300 python tools/fuzz_membrane.py 100 62 2>&1 | tail -1
timeout 300 python tools/fuzz_membrane_frame.py 40 63 2>&1 | tail -1
for S in 1 4 1 4; do
  timeout 300 python bench.py --workload membrane --streams $S --steps 512 --warmup 32 2>/dev/null | python -c "import json,sys;l=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('streams',l['config'].get('streams'),l['value'],l['ms_per_step'],l.get('verify'))"
done
} </dev/null 2>&1 | tee $O/${T}_c5.txt
