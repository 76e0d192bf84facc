#!/bin/bash
# AddressSanitizer + UBSan over the HOST side of libngsc_hip.so (engine / container / decode / multi .cpp; the device code is
# compiled as always - the pool has no GPU sanitizer), on the CPU: the C-ABI tests of the CPU suite and tools/fuzz_host_parsers.py
# against the instrumented build.  Needs the ordinary build first (the kernels' objects are reused).
#   tools/asan_host.sh [rounds]        -> /tmp/ngsc_asan/libngsc_hip.so, report on stdout
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${NGSC_ASAN_DIR:-/tmp/ngsc_asan}
ROUNDS=${1:-20000}
SAN="-fsanitize=address,undefined -fno-gpu-sanitize"
mkdir -p "$OUT"
cd "$ROOT/phyngsc_amd/csrc"
for f in engine container decode multi; do
    $HIPCC --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC $SAN -fno-omit-frame-pointer -c $f.cpp -o "$OUT/$f.o" &
done
wait
$HIPCC --offload-arch=gfx950 -shared $SAN -o "$OUT/libngsc_hip.so" kernels_*.o "$OUT"/engine.o "$OUT"/container.o "$OUT"/decode.o "$OUT"/multi.o -ldl -lpthread
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
cd "$ROOT"
export LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 NGSC_LIB_PATH="$OUT/libngsc_hip.so"
python -m pytest tests/test_cabi_container.py tests/test_distributed_cpu.py -x -q -m "not gpu"
python tools/fuzz_host_parsers.py --n "$ROUNDS"
