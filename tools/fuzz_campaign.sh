mkdir -p gpurun_out/r02/fuzz_final
for s in 501 502 503; do timeout 400 python tools/fuzz_search.py 1500 $s 2>&1 | tail -2; done | tee gpurun_out/r02/fuzz_final/search.log
timeout 300 python tools/fuzz_pipeline.py 400 504 2>&1 | tail -2 | tee gpurun_out/r02/fuzz_final/pipeline.log
timeout 200 python tools/fuzz_membrane.py 100 505 2>&1 | tail -2 | tee gpurun_out/r02/fuzz_final/membrane.log
timeout 200 python tools/fuzz_lipid_order.py 300 506 2>&1 | tail -2 | tee gpurun_out/r02/fuzz_final/lipid.log
timeout 200 python tools/fuzz_xtc.py 100 507 2>&1 | tail -2 | tee gpurun_out/r02/fuzz_final/xtc.log
timeout 300 python tools/fuzz_search_large.py 2>&1 | tail -3 | tee gpurun_out/r02/fuzz_final/large.log
