# A/B of the C5 leg on one box: the regular library and a variant (molar_amd/_ab/libmolar_hip_NAME.so) alternate, one and four contexts,
# plus a bit-for-bit comparison of every membrane output of the two (tools/ab_membrane_fit.py).   usage: tools/ab_c5.sh NAME [rounds]
R=/root/repo; V=${1:-prev}; N=${2:-3}; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
{
MOLAR_HIP_PLUGIN=$R/molar_amd/_ab/libmolar_hip_$V.so timeout 600 python tools/ab_membrane_fit.py dump /tmp/ab_$V.npz 2>&1 | tail -1
timeout 600 python tools/ab_membrane_fit.py dump /tmp/ab_regular.npz 2>&1 | tail -1
python tools/ab_membrane_fit.py compare /tmp/ab_$V.npz /tmp/ab_regular.npz 2>&1 | tail -2
for i in $(seq $N); do for L in $V regular; do for S in 1 4; do
  if [ $L = regular ]; then unset MOLAR_HIP_PLUGIN; else export MOLAR_HIP_PLUGIN=$R/molar_amd/_ab/libmolar_hip_$L.so; fi
  timeout 300 python bench.py --workload membrane --streams $S --steps 512 --warmup 32 2>/dev/null | python -c "import json,sys;l=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$L','streams',$S,round(l['value'],1))"
done; done; done
} </dev/null 2>&1 | tee $O/ab_c5_$V.txt
