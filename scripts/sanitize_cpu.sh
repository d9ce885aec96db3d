#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU code (the C++ host library and the test oracle), driven by the CPU
# test suite.  GPU sanitizers are not available on the pool; this is the CPU half.  Leaves the regular libraries in place.
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer"
for f in param sky image volume_io camera capi; do
  g++ -O1 -g -std=c++17 -fPIC -fopenmp -ffp-contract=off $SAN -Icuda-volpath_amd/host -c cuda-volpath_amd/host/$f.cpp -o $T/$f.o
done
g++ -shared -fPIC -fopenmp $SAN -o $T/libvolpath_host.so $T/*.o
gcc -O1 -g -std=c11 -fPIC -fopenmp -ffp-contract=off -fno-fast-math -fno-math-errno -march=x86-64-v3 $SAN -shared -o $T/libvp_oracle.so oracle/vp_oracle.c -lm
cp cuda-volpath_amd/libvolpath_host.so $T/host.orig; cp oracle/libvp_oracle.so $T/oracle.orig
restore() { cp $T/host.orig cuda-volpath_amd/libvolpath_host.so; cp $T/oracle.orig oracle/libvp_oracle.so; rm -rf $T; }
trap restore EXIT
cp $T/libvolpath_host.so cuda-volpath_amd/libvolpath_host.so; cp $T/libvp_oracle.so oracle/libvp_oracle.so
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
python -m pytest tests/test_host_cpu.py tests/test_oracle_cpu.py tests/test_dist_cpu.py -x -q -m "not gpu" -p no:cacheprovider
