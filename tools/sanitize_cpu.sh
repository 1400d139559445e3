#!/bin/bash
# The host side of the library (builder, file formats, C ABI plumbing, records expander, synthetic generator) and the CLI under AddressSanitizer +
# UndefinedBehaviorSanitizer: the .cpp files rebuilt with -fsanitize=address,undefined and linked with the ordinary device objects, then the whole
# CPU suite (-m "not gpu") with the sanitizer runtime preloaded and output capture off.  (GPU sanitizers are not available on the pool.)
# Restores the product build afterwards.  usage: tools/sanitize_cpu.sh   (from the repo root; needs a built tree)
set -e
ROOT=$(pwd); T=${TMPDIR:-/tmp}/finito_asan; mkdir -p $T
cp finito_amd/libfinito_amd.so $T/lib_good.so; cp finito_amd/finito $T/finito_good
cd finito_amd/csrc
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -fPIC -fopenmp -Wall -Wextra -Wno-unused-parameter -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
for f in fin_build fin_sdsl fin_stats fin_capi fin_synth; do g++ $SAN -c -o $T/$f.o $f.cpp & done; wait
g++ -shared -fsanitize=address,undefined -o ../libfinito_amd.so $T/fin_build.o $T/fin_sdsl.o $T/fin_stats.o $T/fin_capi.o $T/fin_synth.o fin_kernels.o fin_kernel_v2.o fin_kernel_v3.o fin_kernel_w.o fin_kernel_b.o \
    fin_prepass.o fin_build_gpu.o fin_text.o fin_pack.o fin_records.o -L/opt/rocm/lib -lamdhip64 -fopenmp -Wl,-rpath,/opt/rocm/lib
g++ $SAN -o ../finito main.cpp -L.. -lfinito_amd -lz -Wl,-rpath,'$ORIGIN' -L/opt/rocm/lib -lamdhip64 -fopenmp -Wl,-rpath,/opt/rocm/lib
cd $ROOT
set +e
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest tests -x -q -s -m "not gpu" -p no:cacheprovider > $T/pytest.log 2>&1
tail -1 $T/pytest.log
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' $T/pytest.log)"
grep 'runtime error\|AddressSanitizer' $T/pytest.log | sed 's/0x[0-9a-f]*/ADDR/g' | sort | uniq -c | sort -rn | head -20
cp $T/lib_good.so finito_amd/libfinito_amd.so; cp $T/finito_good finito_amd/finito
