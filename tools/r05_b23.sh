O=gpurun_out/r05/b23
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_search.py -x -q -k "misaligned" 2>&1 | tail -4 > $O/tests.txt
cat $O/tests.txt
ROUND=r05 bash tools/fuzz_campaign.sh 2 > $O/fuzz.txt 2>&1
grep -h "cases\|fail\|FAIL\|MISMATCH" $O/fuzz.txt | tail -30
