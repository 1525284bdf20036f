#!/bin/bash
for cfg in "4 4 16384" "5 5 16384" "6 6 16384" "4 3 16384" "5 4 8192" "6 5 8192"; do
  for k in rows columns; do
    FF_ELOC_KERNEL=$k timeout 120 python tools/probes/eloc_ab.py $cfg 2>&1 | grep -v amdgpu | sed "s/steps hist.*rej/rej/" | cut -c1-150
  done
done
