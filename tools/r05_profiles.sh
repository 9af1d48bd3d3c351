# The round's profile set (raw output under gpurun_out/r05; tools/refresh_profiles.py + the copies below make profiles/r05_*)
export ROUND=r05
R=/root/repo
O=$R/gpurun_out/r05
mkdir -p $O
bash $R/tools/profile_bench.sh > $O/profile_bench.log 2>&1
bash $R/tools/pmc_pair_kernels.sh > $O/sq_pair.txt 2>&1
bash $R/tools/pmc_hist_kernel.sh > $O/sq_hist.txt 2>&1
cd $R
MOLAR_HIP_DEBUG_LAUNCH=8 MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_dbg.so python tools/hist_wave_times.py 2>/dev/null > $O/hist_wave_times.json
python bench.py --workload membrane --steps 200 --warmup 10 --verify 2>/dev/null | tail -1 > $O/membrane_bench.json
python bench.py --workload membrane --steps 200 --warmup 10 --streams 1 2>/dev/null | tail -1 > $O/membrane_bench_1ctx.json
python tools/bench_configs.py 2>/dev/null > $O/bench_configs.jsonl
python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>/dev/null
bash tools/r04_timeline.sh r05 > $O/timeline.txt 2>&1
# what comes back through gpurun_out/ is capped at 64 MiB: the per-dispatch traces are not needed once the summaries exist
find $R/gpurun_out -name "*kernel_trace.csv" -delete; find $R/gpurun_out -name "*.db" -delete; find $R/gpurun_out -name "*_agent_info.csv" -delete
du -sh $R/gpurun_out | tail -1
tail -n 3 $O/profile_bench.log; cut -c1-300 $O/membrane_bench.json; cut -c1-200 $O/bench_steps20.json
