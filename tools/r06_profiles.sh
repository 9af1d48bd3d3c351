# The round's profile set (raw output under gpurun_out/r06; tools/refresh_profiles.py + the copies at the end make profiles/r06_*)
export ROUND=r06
R=/root/repo
O=$R/gpurun_out/r06
mkdir -p $O
bash $R/tools/profile_bench.sh > $O/profile_bench.log 2>&1
bash $R/tools/pmc_pair_kernels.sh > $O/sq_pair.txt 2>&1
bash $R/tools/pmc_hist_kernel.sh > $O/sq_hist.txt 2>&1
cd $R
python bench.py --workload rdf --source xtc --steps 512 --warmup 32 --xtc-window 16 --verify 2>/dev/null | tail -1 > $O/rdf_xtc.jsonl
python bench.py --workload rdf --source xtc --steps 512 --warmup 32 --xtc-window 16 --rdf-single-calls 2>/dev/null | tail -1 >> $O/rdf_xtc.jsonl
python bench.py --workload rdf --steps 1024 --warmup 64 --rdf-single-calls --no-cpu-baseline 2>/dev/null | tail -1 > $O/rdf_bench_single_calls.json
python bench.py --workload membrane --steps 512 --warmup 16 --verify 2>/dev/null | tail -1 > $O/membrane_bench.json
python bench.py --workload membrane --steps 256 --warmup 16 --streams 1 2>/dev/null | tail -1 >> $O/membrane_bench.json
python tools/bench_configs.py 2>/dev/null > $O/bench_configs.jsonl
python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>/dev/null
bash tools/r04_timeline.sh r06 > $O/timeline.txt 2>&1; cp $R/gpurun_out/r04/r06_timeline.txt $O/frame_timeline.txt 2>/dev/null
bash tools/r06_rdf_trace.sh final > /dev/null 2>&1
bash tools/r06_membrane_trace.sh final > /dev/null 2>&1
bash tools/r06_c5_timeline.sh final > /dev/null 2>&1
timeout 900 python tools/bench_cutoff_sweep.py 0.3 0.35 0.4 0.5 0.6 0.8 1.0 1.2 1.3 1.4 1.5 1.6 1.8 2.0 2.2 2.4 > $O/cutoff_sweep.jsonl 2>/dev/null
# what comes back through gpurun_out/ is capped at 64 MiB: keep the summaries (kernel statistics, the FETCH / WRITE counter tables of the
# three-step passes, text and JSON), drop the per-dispatch traces and every other raw file of rocprofv3
find $R/gpurun_out -name "*kernel_trace.csv" -delete; find $R/gpurun_out -name "*.db" -delete; find $R/gpurun_out -name "*_agent_info.csv" -delete
find $R/gpurun_out -type f -name "*counter_collection.csv" ! -path "*/r06/fetch/*" ! -path "*/r06/write/*" -delete
find $R/gpurun_out -type f ! -name "*.txt" ! -name "*.json" ! -name "*.jsonl" ! -name "*.csv" ! -name "*.log" ! -name "*.err" -delete
find $R/gpurun_out -type f -size +20M -delete
du -sh $R/gpurun_out/* $R/gpurun_out/r06/* 2>/dev/null | sort -h | tail -8
du -sh $R/gpurun_out | tail -1
tail -n 3 $O/profile_bench.log; cut -c1-300 $O/membrane_bench.json; cut -c1-200 $O/bench_steps20.json
