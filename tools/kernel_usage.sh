#!/bin/bash
# Register / scratch / LDS usage of ONE instantiation of a local-energy kernel, in seconds (hipcc cross-compiles without a GPU):
#   tools/kernel_usage.sh rows N SPLIT [WPS] [extra hipcc flags]      ff_eloc_rows_kernel<N, 2, SPLIT, true, WPS>
#   tools/kernel_usage.sh mfma N [WPS] [extra hipcc flags]            ff_eloc_mfma_kernel<N, 2, true, WPS>
# (a translation unit with the shared prologue of ff_cnf_fwd.hip and that single explicit instantiation is written to /tmp)
set -e
kind=$1; shift
csrc=$(cd "$(dirname "$0")/../fermiflow_amd/csrc" && pwd)
python3 - "$csrc" "$kind" "$@" <<'PY'
import sys
csrc, kind, args = sys.argv[1], sys.argv[2], sys.argv[3:]
src = open(csrc + "/ff_cnf_fwd.hip").read()
head = src[:src.index("template <int N, int D, int MODE, bool TAB>\n__global__ void __launch_bounds__(FF_WAVE,")]
if kind == "rows":
    n, split = args[0], args[1]; wps = args[2] if len(args) > 2 and args[2].isdigit() else "1"
    inst = f'#include "ff_eloc_rows.h"\ntemplate __global__ void ff_eloc_rows_kernel<{n}, 2, {split}, true, {wps}>(ff_fwd_args);\n'
else:
    n = args[0]; wps = args[1] if len(args) > 1 and args[1].isdigit() else "1"
    inst = f'#include "ff_eloc_mfma.h"\ntemplate __global__ void ff_eloc_mfma_kernel<{n}, 2, true, {wps}>(ff_fwd_args);\n'
open("/tmp/ff_kernel_usage.hip", "w").write(head + inst)
PY
extra=()
for a in "$@"; do case "$a" in -*) extra+=("$a");; esac; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -mllvm -disable-machine-licm --cuda-device-only -I"$csrc" "${extra[@]}" \
  -c /tmp/ff_kernel_usage.hip -o /tmp/ff_kernel_usage.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "VGPRs:|AGPRs|Scratch|LDS Size|Occupancy" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' | paste - - - - - | tail -1
