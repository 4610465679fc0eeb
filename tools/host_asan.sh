#!/bin/bash
# Host C code under AddressSanitizer + UBSan (CPU build only: the GPU pool has no device ASAN).
# Rebuilds host/*.c with gcc -fsanitize=address,undefined, links them with the ordinary HIP objects
# into gpurun_out/asan/libpll_amd_asan.so and runs the CPU test suite against that library.
# Usage: tools/host_asan.sh            (after `make -C libpll-2_amd/csrc`)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/libpll-2_amd/csrc
OUT=$ROOT/gpurun_out/asan
mkdir -p "$OUT"
for f in "$SRC"/host/*.c; do
  gcc -O1 -g -std=gnu11 -fPIC -D_GNU_SOURCE -fsanitize=address,undefined -fno-omit-frame-pointer \
      -c "$f" -o "$OUT/$(basename "${f%.c}").o"
done
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$OUT/libpll_amd_asan.so" "$OUT"/*.o \
    "$SRC/hip/pllgpu.o" "$SRC/hip/compress.o" -lm "$ASAN" "$UBSAN" -Wl,-rpath,/opt/rocm/lib
cd "$ROOT"
# the loader's RTLD_DEEPBIND is refused by the sanitizer runtime; python itself leaks by design
LD_PRELOAD=$ASAN PLL_AMD_NO_DEEPBIND=1 PLL_AMD_LIB=$OUT/libpll_amd_asan.so \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest tests -q -m "not gpu" --no-header -p no:cacheprovider 2>&1 | tee "$OUT/run.log"
echo "UBSan reports: $(grep -c 'runtime error' "$OUT/run.log" || true)"
