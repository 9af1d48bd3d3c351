O=gpurun_out/r05/b15
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_search.py -x -q -k "connectivity or unwrap" 2>&1 | tail -5 > $O/tests.txt
timeout 900 python -m pytest tests/test_cpp_host.py -x -q 2>&1 | tail -3 >> $O/tests.txt
cat $O/tests.txt
