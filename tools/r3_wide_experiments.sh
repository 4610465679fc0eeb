#!/bin/bash
# what bounds k_partials_mfma_wide? The same C5 launches with parts of the work taken away (results are wrong in every
# variant: measurement only). Builds four extra copies of the library with -DWIDE_EXPERIMENT=n.
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r3h"; rm -rf "$O"; mkdir -p "$O"
cd "$R/libpll-2_amd/csrc"
for e in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DWIDE_EXPERIMENT=$e -c hip/pllgpu.hip -o /tmp/pllgpu_e$e.o 2>/dev/null &
done
wait
for e in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /tmp/libpll_e$e.so host/*.o /tmp/pllgpu_e$e.o hip/compress.o -lm -ldl -lrt -Wl,-rpath,/opt/rocm/lib
done
cd "$R"
export PLL_BENCH_MEASUREMENT_BUILD=1
for e in 0 1 2 3 4 0; do
  if [ $e = 0 ]; then unset PLL_AMD_LIB; else export PLL_AMD_LIB=/tmp/libpll_e$e.so; fi
  python3 bench.py --config c5 --steps 10 --blocks 3 --no-cpu > "$O/c5_e${e}_$RANDOM.json" 2>> "$O/c5.err"
done
python3 - <<'PY'
import json,glob,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3h"
names={"0":"as shipped","1":"children from 64 tiles (cache hits)","2":"parent stores dropped","3":"both","4":"half of the left child's MFMAs skipped (-25 %)"}
for f in sorted(glob.glob(O+"/c5_e*.json"), key=os.path.getmtime):
    try: d=json.load(open(f))
    except Exception: continue
    r=d["roofline"]; e=os.path.basename(f)[4]
    print("%-36s avg ii launch %.1f us  step %.1f us" % (names[e], r["avg_launch_ms"]*1e3, d["ms_per_step"]*1e3))
PY
