#!/bin/bash
for cfg in "2 0 6 0" "2 0 6 1" "2 0 17 0" "3 0 6 0" "4 0 17 1" "5 0 6 0" "5 0 17 1" "5 0 33 0" "5 0 64 0" "3 3 6 0" "3 3 17 1" "3 3 33 0" "4 4 7 0" "5 5 5 1" "6 6 3 0"; do
  timeout 60 python tools/probes/dbg_n2.py $cfg 2>&1 | grep -v amdgpu.ids | tail -2
done
