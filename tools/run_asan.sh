#!/bin/bash
# CPU-side sanitizer run (SURVEY.md 5 "race detection / sanitizers"): the host layer of libscema_md.so
# (host/*.cpp: LAMMPS restart / data parsers, ReaxFF parameter reader, FlatJson, STMDSync, continuum stand-in, clustering), the CPU
# oracles (oracle/*.c) and the ReaxFF host driver (tests/reax_host_driver.cpp) under AddressSanitizer + UndefinedBehaviorSanitizer.
# The sanitizer runtimes are preloaded into python; SCEMA_SANITIZE=1 makes the loaders take the instrumented libraries.  No GPU.
set -e
cd "$(dirname "$0")/.."
make -C scema_amd/csrc -j8 -s asan
make -C oracle -s asan
export SCEMA_SANITIZE=1
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1:allocator_may_return_null=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)${LD_PRELOAD:+:$LD_PRELOAD}"
exec python -m pytest -x -q -p no:cacheprovider -m "not gpu" \
  tests/test_corrupt_files.py tests/test_formats.py tests/test_host_arithmetic.py tests/test_reax_host.py tests/test_stmd_sync_host.py \
  tests/test_cluster.py tests/test_oracle_physics.py tests/test_oracle_reax.py "$@"
