O=gpurun_out/r05/b7
mkdir -p $O
for v in dbg dbgsync dbgwgs; do
MOLAR_HIP_DEBUG_LAUNCH=8 MOLAR_HIP_PLUGIN=molar_amd/_ab/libmolar_hip_$v.so python tools/hist_wave_times.py 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$v', 'span', d['kernel_span_us'], 'life', d['wave_life_us_mean'], 'idle_end', d['share_of_span']['idle_at_end'], 'pre', d['us_per_slot']['preamble'], 'longest', d['longest_slot_us'], 'top3', [(x['us'], x['flags'], x['nch'], x['ticket_wait_us']) for x in d['longest_slots'][:3]])
" >> $O/sum.txt
done
bash tools/ab_rdf.sh molar_amd/_ab/libmolar_hip_dbg.so molar_amd/_ab/libmolar_hip_dbgsync.so molar_amd/_ab/libmolar_hip_dbgwgs.so >> $O/sum.txt 2>&1
cat $O/sum.txt
