O=gpurun_out/r05/b13
mkdir -p $O
python bench.py --workload rdf --source xtc --steps 200 --warmup 10 --verify > $O/rdf_xtc.json 2> $O/rdf_xtc.err
python bench.py --workload rdf --source xtc --decoder device --steps 2048 --warmup 10 --verify > $O/rdf_xtc_dev.json 2> $O/rdf_xtc_dev.err
python bench.py --gpus 2 --share-gpu --workload rdf --source xtc --steps 100 --warmup 10 --verify > $O/rdf_xtc_2.json 2> $O/rdf_xtc_2.err
tail -n 2 $O/*.json; tail -n 3 $O/*.err
