#!/bin/bash
# The host helper (polyfuzz_amd/csrc_host/_pack.c: the crews of the string packer and of the frame's range fill) under
# AddressSanitizer and ThreadSanitizer on the CPU build -- GPU sanitizers are not available on the pool; the crews' lifetime rules
# (detached helpers, a job freed by whoever lets go last, phases that end on work counts) are what these two check.
# usage: bash tools/host_sanitizers.sh      (restores the shipped _pack.so afterwards)
cd "$(dirname "$0")/.."
INC=$(python -c "import sysconfig; print(sysconfig.get_paths()['include'])")
cp polyfuzz_amd/_pack.so /tmp/_pack_shipped.so
for san in address thread; do
  gcc -O1 -g -fsanitize=$san -fno-omit-frame-pointer -msse4.1 -shared -fPIC -I "$INC" polyfuzz_amd/csrc_host/_pack.c -lm -lpthread -o polyfuzz_amd/_pack.so || break
  lib=$(gcc -print-file-name=lib$( [ $san = address ] && echo asan || echo tsan ).so)
  rm -f /tmp/${san}_log.*
  ASAN_OPTIONS=detect_leaks=0:log_path=/tmp/address_log TSAN_OPTIONS=report_signal_unsafe=0:halt_on_error=0:log_path=/tmp/thread_log \
    LD_PRELOAD=$lib timeout 1500 python -m pytest tests/test_frame_ranges_cpu.py tests/test_host_logic_cpu.py -x -q 2>&1 | tail -1
  echo "$san sanitizer reports: $(ls /tmp/${san}_log.* 2>/dev/null | wc -l)"; cat /tmp/${san}_log.* 2>/dev/null | grep "SUMMARY" | sort | uniq -c
done
cp /tmp/_pack_shipped.so polyfuzz_amd/_pack.so
