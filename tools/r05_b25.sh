O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_search.py -x -q -k "interleaved or pairs_plane or misaligned" 2>&1 | tail -3 > $O/b25_tests.txt
bash tools/pmc_icache.sh > $O/icache.txt 2>&1
cat $O/b25_tests.txt; grep -v amdgpu $O/icache.txt
