#!/bin/bash
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r3b"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma61_probe.hip -o /tmp/mfma61_probe > /dev/null 2>&1 && /tmp/mfma61_probe > "$O/mfma61_probe.txt" 2>&1
cat "$O/mfma61_probe.txt"
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$O/pytest.log"
tail -15 "$O/pytest.log"
