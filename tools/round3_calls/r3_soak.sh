#!/bin/bash
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/soak"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout -k 10 600 python3 tools/soak.py 300 > "$O/soak.txt" 2>&1; echo "soak rc=$?" >> "$O/soak.txt"; tail -8 "$O/soak.txt"
