cd /root/repo
timeout 900 python tools/fuzz_hist_frames.py 200 31337 2>&1 | tail -2
timeout 900 python tools/fuzz_slab.py 200 31338 2>&1 | tail -2
timeout 900 python tools/fuzz_membrane_frame.py 120 31339 2>&1 | tail -1
timeout 900 python tools/fuzz_search.py 4000 31340 2>&1 | tail -1
timeout 600 python tools/fuzz_lipid_order.py 1500 31341 2>&1 | tail -1
timeout 600 python tools/fuzz_membrane.py 300 31342 2>&1 | tail -1
timeout 600 python tools/fuzz_xtc.py 300 31343 2>&1 | tail -1
