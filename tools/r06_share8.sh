# Functional runs of the N > 1 code path of all bench workloads with EIGHT ranks sharing the box's one GPU (--share-gpu: rank r uses
# GPU r mod count, reductions over gloo).  NOT a scaling result: eight processes time-slice one chip.  What it shows: eight ranks
# launch, shard their frames, keep exact results (each line's self-check) and fit in memory and host threads together.
# usage: tools/r06_share8.sh   -> gpurun_out/r06/share8_<workload>.json (+ peak VRAM in share8_vram.txt)
R=/root/repo; O=$R/gpurun_out/r06; mkdir -p $O
run() {   # tag, bench flags
    tag=$1; shift
    ( while true; do rocm-smi --showmeminfo vram 2>/dev/null | grep "Used Memory" ; sleep 2; done ) > $O/share8_${tag}_vram.log 2>&1 &
    mon=$!
    s=$(date +%s)
    timeout 900 python $R/bench.py --gpus 8 --share-gpu "$@" > $O/share8_$tag.json 2> $O/share8_$tag.err
    rc=$?
    kill $mon 2>/dev/null
    peak=$(grep -o "[0-9]*$" $O/share8_${tag}_vram.log | sort -n | tail -1)
    echo "$tag rc=$rc seconds=$(( $(date +%s) - s )) peak_vram_bytes=$peak" | tee -a $O/share8_vram.txt
    rm -f $O/share8_${tag}_vram.log
}
rm -f $O/share8_vram.txt
run search_fit --steps 16 --warmup 4 --no-cpu-baseline
run rdf --workload rdf --steps 64 --warmup 16 --verify
run rdf_xtc --workload rdf --source xtc --steps 64 --warmup 16 --verify
run membrane --workload membrane --steps 32 --warmup 8 --verify
