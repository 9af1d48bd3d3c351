# Builds a variant of the library for A/B runs on one box: tools/build_variant.sh NAME "-DFLAG ..." [sources...]
# Recompiles the given translation units (default: pair_k0.hip; "all" = every unit) with the extra flags, in parallel, and
# links them with the objects of the regular build into molar_amd/_ab/libmolar_hip_NAME.so (select with MOLAR_HIP_PLUGIN).
# Flags that change a struct shared between units (-DMOLAR_HIP_DEBUG_KNOBS: SearchParams) need "all".
set -e
name=$1; flags=$2; shift 2
srcs=${@:-pair_k0.hip}
cd "$(dirname "$0")/../molar_amd"
mkdir -p _ab
all="api search search_f64 measure measure_f64 membrane xtc pair_k0 pair_k1 pair_k2 pair_k3 pair_k4 pair_k5 pair_k6 pair_k7 pair_k8 pair_small devsort"
if [ "$srcs" = all ]; then srcs=$(for f in $all; do printf "%s.hip " $f; done); fi
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -fno-slp-vectorize"
objs=""
pids=""
for f in $all; do
  if echo " $srcs " | grep -q " $f.hip "; then
    /opt/rocm/bin/hipcc $base $flags -c csrc/$f.hip -o _ab/${f}_$name.o &
    pids="$pids $!"
    objs="$objs _ab/${f}_$name.o"
  elif [ -f _ab/${f}_$name.o ] && [ "$REUSE" = 1 ]; then
    objs="$objs _ab/${f}_$name.o"          # REUSE=1: units of this variant compiled by an earlier call (flags that change shared structs)
  else
    objs="$objs csrc/$f.o"
  fi
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/libmolar_hip_$name.so $objs
echo _ab/libmolar_hip_$name.so
