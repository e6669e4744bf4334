#!/bin/bash
# ASan + UBSan run of the CPU-side code (the reference has no sanitizer story; this is ours).
#   1. the oracle (oracle/oracle_imgdecode.c) and the synthetic encoder (oracle/jpeg_synth.c), built with
#      gcc -fsanitize=address,undefined and driven by the oracle test suites and a bounded fuzz campaign;
#   2. the host library's device-free paths -- JFIF front end, decode-table builders and their self test, geometry / descriptor
#      code -- built with hipcc's clang -fsanitize=address,undefined (host only) and driven by tools/sanitize/host_paths.cpp.
# Writes tools/sanitize/last_run.txt; any sanitizer report makes the script fail.
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=tools/sanitize/last_run.txt
: > $OUT
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g"
fail=0

echo "== 1. oracle + synth under gcc ASan/UBSan (alloc_dealloc_mismatch off: the compiled REFERENCE frees a new[] block with delete in ~CwindowBuf, source/WindowBuf.cpp; not our code)" | tee -a $OUT
mkdir -p /tmp/jsnoop_san
gcc -O1 -ffp-contract=off -fPIC -std=c11 $SAN -shared -o /tmp/jsnoop_san/liboracle_imgdecode.so oracle/oracle_imgdecode.c -lm || fail=1
gcc -O1 -ffp-contract=off -fPIC -std=c11 $SAN -shared -o /tmp/jsnoop_san/libjsnoop_synth.so oracle/jpeg_synth.c -lm || fail=1
export JSNOOP_ORACLE_DIR=/tmp/jsnoop_san
ASAN_RT=$(gcc -print-file-name=libasan.so)
if LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:alloc_dealloc_mismatch=0:exitcode=23 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
   timeout 1500 python -m pytest tests/test_oracle_golden.py tests/test_oracle_vs_ref.py tests/test_progressive_pillow.py -x -q -m "not gpu" -p no:cacheprovider > /tmp/jsnoop_san/oracle_tests.log 2>&1; then
  tail -1 /tmp/jsnoop_san/oracle_tests.log | tee -a $OUT
else
  echo "FAILED (see below)" | tee -a $OUT; tail -30 /tmp/jsnoop_san/oracle_tests.log | tee -a $OUT; fail=1
fi
if [ -d /root/reference/source ]; then
  if LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:alloc_dealloc_mismatch=0:exitcode=23 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
     timeout 1500 python tools/fuzz_oracle_vs_ref.py 3000 7 > /tmp/jsnoop_san/oracle_fuzz.log 2>&1; then
    echo "fuzz_oracle_vs_ref.py 3000 cases: $(tail -1 /tmp/jsnoop_san/oracle_fuzz.log)" | tee -a $OUT
  else
    echo "fuzz FAILED" | tee -a $OUT; tail -30 /tmp/jsnoop_san/oracle_fuzz.log | tee -a $OUT; fail=1
  fi
fi
unset JSNOOP_ORACLE_DIR

echo "== 2. host library, device-free paths, under clang ASan/UBSan" | tee -a $OUT
HIPCC=/opt/rocm/bin/hipcc
SRC=jpegsnoop_amd/csrc
if $HIPCC --offload-arch=gfx950 -O1 -std=c++17 -ffp-contract=off -x hip $SAN -fno-gpu-sanitize \
     -o /tmp/jsnoop_san/host_paths tools/sanitize/host_paths.cpp $SRC/jfif_front.cpp $SRC/jsnoop_parallel.cpp $SRC/jsnoop_host.cpp $SRC/jsnoop_report.cpp \
     $SRC/jsnoop_tiff.cpp $SRC/jsnoop_progressive.cpp $SRC/jsnoop_pipeline.cpp $SRC/jsnoop_kernels.hip $SRC/jsnoop_progressive.hip \
     -L/opt/rocm/lib -lrocprofiler-sdk-roctx -Wl,-rpath,/opt/rocm/lib > /tmp/jsnoop_san/host_build.log 2>&1; then
  if ASAN_OPTIONS=detect_leaks=0:exitcode=23 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 timeout 900 /tmp/jsnoop_san/host_paths oracle/libjsnoop_synth.so > /tmp/jsnoop_san/host_run.log 2>&1; then
    tail -3 /tmp/jsnoop_san/host_run.log | tee -a $OUT
  else
    echo "host_paths FAILED" | tee -a $OUT; tail -30 /tmp/jsnoop_san/host_run.log | tee -a $OUT; fail=1
  fi
else
  echo "host build FAILED" | tee -a $OUT; tail -20 /tmp/jsnoop_san/host_build.log | tee -a $OUT; fail=1
fi
echo "== result: $([ $fail = 0 ] && echo clean || echo FAILED)" | tee -a $OUT
exit $fail
