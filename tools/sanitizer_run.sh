#!/bin/bash
# The library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer (the reference runs none).
#   tools/sanitizer_run.sh [pytest arguments ...]        default: the CPU suite (-m "not gpu")
# Works on a copy of the tree in /tmp with wayverb_amd/sanitized/libwayverb_amd.so (python -m wayverb_amd.build --sanitized;
# device code as shipped) in the product library's place, the ASan runtime preloaded into python.  Leak checking is off
# (the interpreter and the HIP runtime never free what they hold at exit); an ASan report aborts the process it happens in,
# UBSan reports are collected (all of them, every process) and listed at the end: any report fails the run.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
[ -f "$ROOT/wayverb_amd/sanitized/libwayverb_amd.so" ] || (cd "$ROOT" && python -m wayverb_amd.build --sanitized > /dev/null)
COPY=/tmp/wayverb_amd_sanitized_tree
rm -rf "$COPY"; mkdir -p "$COPY"
(cd "$ROOT" && tar --exclude=.git --exclude=gpurun_out --exclude='wayverb_amd/sanitized/*.o' -cf - .) | tar -xf - -C "$COPY"
cp "$COPY/wayverb_amd/sanitized/libwayverb_amd.so" "$COPY/wayverb_amd/libwayverb_amd.so"
touch "$COPY/wayverb_amd/libwayverb_amd.so" "$COPY/wayverb_amd/csrc/engine.resources.txt"   # newer than the sources: build() leaves it alone
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd "$COPY"
if [ $# -eq 0 ]; then set -- tests -q -x -m "not gpu"; fi
# (ASan's dlopen interceptor hides the caller from the loader, so a library's own RUNPATH no longer finds its neighbours:
# torch's directory goes on the search path by hand)
TORCH_LIB=$(python -c "import importlib.util as u; print(u.find_spec('torch').submodule_search_locations[0] + '/lib')")
rm -f /tmp/wv_ubsan.log.*
# (quarantine: large enough that nothing is recycled while the HIP runtime unloads at exit -- ROCm's ASan runtime trips over
# its own device allocator there, sanitizer_allocator_device.h:125, when a process has freed more than the default 256 MB)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:protect_shadow_gap=0:halt_on_error=1:quarantine_size_mb=16384
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=/tmp/wv_ubsan.log
set +e
LD_PRELOAD="$RT" LD_LIBRARY_PATH="$(dirname "$RT"):$TORCH_LIB:$LD_LIBRARY_PATH" python -m pytest "$@"
rc=$?
reports=$(cat /tmp/wv_ubsan.log.* 2>/dev/null | grep -c "runtime error")
echo "UBSan reports: $reports"
cat /tmp/wv_ubsan.log.* 2>/dev/null | grep "runtime error" | sort | uniq -c
others=$(cat /tmp/wv_ubsan.log.* 2>/dev/null | grep -c "ERROR: AddressSanitizer")
echo "ASan reports: $others"
[ "$others" -eq 0 ] || cat /tmp/wv_ubsan.log.* | head -150
[ "$rc" -eq 0 ] && [ "$reports" -eq 0 ] && [ "$others" -eq 0 ]
