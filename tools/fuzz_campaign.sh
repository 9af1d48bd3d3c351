# Differential fuzz campaign of every entry family against the oracle on one GPU box (about 12 minutes with ROUNDS=10).
# Usage: bash tools/fuzz_campaign.sh [ROUNDS]; logs under gpurun_out/$ROUND/fuzz_final/
T=${ROUND:-r02}
N=${1:-1}
O=gpurun_out/$T/fuzz_final
mkdir -p $O
for r in $(seq 1 $N); do
  s=$((1000 * r))
  timeout 600 python tools/fuzz_search.py 1500 $((s + 1)) 2>&1 | tail -1 | tee -a $O/search.log
  timeout 600 python tools/fuzz_search.py 1500 $((s + 2)) 2>&1 | tail -1 | tee -a $O/search.log
  timeout 300 python tools/fuzz_pipeline.py 400 $((s + 3)) 2>&1 | tail -1 | tee -a $O/pipeline.log
  timeout 400 python tools/fuzz_hist_frames.py 60 $((s + 12)) 2>&1 | tail -1 | tee -a $O/hist_frames.log
  timeout 600 python tools/fuzz_slab.py 60 $((s + 13)) 2>&1 | tail -1 | tee -a $O/slab.log
  timeout 300 python tools/fuzz_membrane.py 100 $((s + 4)) 2>&1 | tail -1 | tee -a $O/membrane.log
  timeout 300 python tools/fuzz_membrane_frame.py 40 $((s + 10)) 2>&1 | tail -1 | tee -a $O/membrane_frame.log
  timeout 300 python tools/fuzz_lipid_order.py 300 $((s + 5)) 2>&1 | tail -1 | tee -a $O/lipid.log
  timeout 300 python tools/fuzz_xtc.py 100 $((s + 6)) 2>&1 | tail -1 | tee -a $O/xtc.log
  timeout 300 python tools/fuzz_fit.py 400 $((s + 7)) 2>&1 | tail -1 | tee -a $O/fit.log
  timeout 300 python tools/fuzz_measure.py 300 $((s + 8)) 2>&1 | tail -1 | tee -a $O/measure.log
  timeout 300 python tools/fuzz_measure_f64.py 300 $((s + 9)) 2>&1 | tail -1 | tee -a $O/measure_f64.log
  timeout 600 python tools/fuzz_search_f64.py 600 $((s + 11)) 2>&1 | tail -1 | tee -a $O/search_f64.log
done
timeout 600 python tools/fuzz_search_large.py 2>&1 | tail -1 | tee -a $O/large.log
