# Builds a variant of the library for A/B runs on one box: tools/build_variant.sh NAME "-DFLAG ..." [sources...]
# Recompiles the given translation units (default: pair_k0.hip) with the extra flags and links them with the objects of
# the regular build into molar_amd/_ab/libmolar_hip_NAME.so (select with MOLAR_HIP_PLUGIN).
set -e
name=$1; flags=$2; shift 2
srcs=${@:-pair_k0.hip}
cd "$(dirname "$0")/../molar_amd"
mkdir -p _ab
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -fno-slp-vectorize"
objs=""
for f in api search search_f64 measure measure_f64 membrane xtc pair_k0 pair_k1 pair_k2 pair_k3 pair_k4 pair_k5; do
  if echo " $srcs " | grep -q " $f.hip "; then
    /opt/rocm/bin/hipcc $base $flags -c csrc/$f.hip -o _ab/${f}_$name.o
    objs="$objs _ab/${f}_$name.o"
  else
    objs="$objs csrc/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/libmolar_hip_$name.so $objs
echo _ab/libmolar_hip_$name.so
