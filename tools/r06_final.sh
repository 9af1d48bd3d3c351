# final pass of the round on one box: the whole GPU suite, the driver's bench command, the profile set
O=gpurun_out/r06
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/final_tests.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/final_bench_driver.json 2> $O/final_bench_driver.err
bash tools/r06_profiles.sh > $O/final_profiles.log 2>&1
cat $O/final_tests.txt; tail -n 4 $O/final_bench_driver.err; cut -c1-250 $O/final_bench_driver.json; tail -n 3 $O/final_profiles.log
