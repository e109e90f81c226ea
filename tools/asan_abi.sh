#!/bin/bash
# CPU-side sanitizer run of the C ABI's host layer: the library is rebuilt with AddressSanitizer and
# UndefinedBehaviorSanitizer on the HOST code only (device-side sanitizers are not available on this pool), and the ABI
# tests (symbol table, struct layout, argument validation; no GPU, no launches) run against it.
#   tools/asan_abi.sh [build-dir]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-$(mktemp -d)}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
RT="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
"$HIPCC" -O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -shared -fsanitize=address,undefined -fno-gpu-sanitize \
    -fno-omit-frame-pointer -shared-libsan -ffp-contract=fast-honor-pragmas \
    "$ROOT"/ship-track-estimators_amd/csrc/*.hip -o "$OUT/libste_hip_asan.so" 2>/dev/null
nm -D "$OUT/libste_hip_asan.so" > "$OUT/symbols.txt"   # (not piped into grep -q: under pipefail that races with SIGPIPE)
grep -q __asan_init "$OUT/symbols.txt"
cd "$ROOT"
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    STE_LIB_PATH="$OUT/libste_hip_asan.so" python -m pytest tests/test_abi.py -x -q -p no:cacheprovider
