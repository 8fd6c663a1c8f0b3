#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (no GPU sanitizer exists on this pool): the oracle's C restatement
# (oracle/gpca_oracle.c -> oracle/_asan/), the native host program's parsers / writers / argument handling (genomic_pca_amd/host ->
# bin/genomic_pca_asan) and the parser harness of tests/cpp/dump_formats.cpp.  Runs the CPU test files against those builds.
#   usage: scripts/sanitize_cpu.sh          (exit code 0 = no sanitizer report, every test green)
set -e
cd "$(dirname "$0")/.."
make -s -C oracle asan
make -s -C genomic_pca_amd/host asan
ASAN_LIB=$(gcc -print-file-name=libasan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
echo "== oracle under ASan + UBSan (tests/test_oracle.py, tests/test_launch.py, tests/test_dist_gloo.py)"
GPCA_ORACLE_SANITIZE=1 LD_PRELOAD=$ASAN_LIB python -m pytest tests/test_oracle.py tests/test_launch.py tests/test_dist_gloo.py -x -q -m "not gpu" -p no:cacheprovider
echo "== host program and parsers under ASan + UBSan (tests/test_cpp_host.py, tests/test_io_cli.py: the CPU tests)"
GPCA_HOST_SANITIZE=1 python -m pytest tests/test_cpp_host.py tests/test_io_cli.py tests/test_abi.py -x -q -m "not gpu" -p no:cacheprovider
echo "sanitize_cpu: clean"
