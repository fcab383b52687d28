#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the HOST side of the library (csrc/api.hip and every launcher: ~3 000 lines of layout, packing and
# dispatch code reachable without a device).  CPU box only -- never on the GPU pool (GPU sanitizers are not available there).
#   tools/sanitize_host.sh        builds build/san/libtepose_hip.so (hipcc -fsanitize=address,undefined -fno-gpu-sanitize: host code instrumented, device code as shipped), the plain-C exerciser
#                                 tests/c_client/tepose_host_check.c against it, runs that, then tests/test_dispatch.py + tests/test_abi.py in a Python that
#                                 has the sanitizer runtime preloaded and TEPOSE_AMD_LIB pointing at the instrumented library.
#   tools/sanitize_host.sh --quick   the build + the C exerciser only (what the CPU test suite runs: ~1 min, most of it the build)
# Exit code 0 = no report.  tests/test_sanitize_host.py runs this script (skipped where the runtime is missing).
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so 2>/dev/null | head -1)
[ -n "$RT" ] || { echo "sanitizer runtime not found"; exit 77; }
mkdir -p build/san
SRC="gemm gemm_h3 gemm_h3s gemm_h3s16c gru_step16 skinny skinny_h3 gru_seq reg_seq misc smpl metrics filters api"
FILES=""; for f in $SRC; do FILES="$FILES tepose_amd/csrc/$f.hip"; done
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -shared-libasan -fno-omit-frame-pointer -g -O1"
if [ ! -f build/san/libtepose_hip.so ] || [ -n "$(find tepose_amd/csrc include -newer build/san/libtepose_hip.so -type f | head -1)" ]; then
  # (device code is compiled as usual and NOT instrumented: -fno-gpu-sanitize; --offload-host-only would leave the fat-binary symbols undefined)
  hipcc --offload-arch=gfx950 -fno-gpu-sanitize -std=c++17 -fPIC -shared $SAN -Xclang -target-feature -Xclang -packed-fp32-ops -DTEPOSE_NO_PACKED_FP32=1 -Wno-inline-asm -Wno-unused-value -o build/san/libtepose_hip.so $FILES
fi
/opt/rocm/lib/llvm/bin/clang $SAN -o build/san/tepose_host_check tests/c_client/tepose_host_check.c -Lbuild/san -ltepose_hip -Wl,-rpath,$ROOT/build/san -Wl,-rpath,$(dirname $RT) -Wl,-rpath,/opt/rocm/lib
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
echo "== plain-C exerciser"
build/san/tepose_host_check
[ "$1" = "--quick" ] && { echo "sanitize_host: no report (quick: C exerciser only)"; exit 0; }
echo "== python tests on the instrumented library"
LD_PRELOAD=$RT TEPOSE_AMD_LIB=$ROOT/build/san/libtepose_hip.so TEPOSE_SANITIZE_RUN=1 python -m pytest tests/test_dispatch.py tests/test_abi.py -x -q -p no:cacheprovider 2>&1 | tail -5
echo "sanitize_host: no report"
