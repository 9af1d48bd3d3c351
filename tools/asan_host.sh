# Host side of libmolar_hip.so under AddressSanitizer + UBSan on a machine WITHOUT a GPU: the entry points that
# run without a device (PeriodicBox helpers, histogram edges, the XTC index and host decoder incl. its handling of
# truncated / corrupt files, patch lists, initial normals, argument validation of everything else) through the CPU tests.
# GPU-side sanitizers are not available on this pool; device code is compiled as usual.
#   bash tools/asan_host.sh            (about 4 minutes on 8 cores; needs a regular build for the pair_k*.o objects)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=${ASAN_OUT:-/tmp/molar_asan}
mkdir -p $O
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SAN="-fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -g"
FLAGS="--offload-arch=gfx950 -O1 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize $SAN"
cd $O
pids=""
for s in api xtc measure membrane search search_f64 measure_f64; do
  $HIPCC $FLAGS -c $R/molar_amd/csrc/$s.hip -o $O/$s.o > $O/$s.log 2>&1 &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
objs="$O/api.o $O/xtc.o $O/measure.o $O/membrane.o $O/search.o $O/search_f64.o $O/measure_f64.o"
for k in 0 1 2 3 4 5 6; do objs="$objs $R/molar_amd/csrc/pair_k$k.o"; done
objs="$objs $R/molar_amd/csrc/pair_small.o $R/molar_amd/csrc/devsort.o"
$HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $O/libmolar_hip.so $objs
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd $R
export MOLAR_HIP_PLUGIN=$O/libmolar_hip.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python -m pytest tests/test_xtc_cpu.py tests/test_abi_cpu.py tests/test_analysis_task_py_cpu.py tests/test_membrane_host_cpu.py -x -q -s -p no:cacheprovider > $O/pytest.log 2>&1 || true
grep -n "runtime error\|AddressSanitizer\|passed\|failed\|Fatal" $O/pytest.log | head -20
# the C++ host mirror (include/molar_hip.hpp: task driver, frame windows, XTC reader) over the same library
unset LD_PRELOAD
CXX=/opt/rocm/lib/llvm/bin/clang++
$CXX -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared-libsan -I $R/include $R/tests/cpp/test_analysis_task.cpp \
  -o $O/test_analysis_task -L $O -lmolar_hip -Wl,-rpath,$O -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$(dirname $RT)
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 $O/test_analysis_task $R/tests/golden/benzene.xtc 2>&1 | tail -3
