#!/bin/bash
# Variant library whose device code went through tools/asm_lds_split.py (LDS double reads taken apart): hipcc's own steps
# (hipcc -### shows them) with the rewrite between the device compiler and the assembler.
#   usage: tools/build_split.sh NAME "ff_cnf_fwd ff_cnf_adj ..." [extra hipcc flags]   -> fermiflow_amd/libfermiflow_hip_NAME.so
# Sources not listed are taken from the regular build's objects (make -C fermiflow_amd/csrc first).
set -e
NAME=$1; SPLIT="$2"; shift 2; EXTRA="$@"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/fermiflow_amd/csrc
LL=/opt/rocm/lib/llvm/bin
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-function -Wno-unused-variable -Wno-pass-failed"
T=${TMPDIR:-/tmp}/ffsplit_$NAME; mkdir -p $T
OBJS=
for f in ff_api ff_comm ff_walkers ff_cnf_fwd ff_cnf_adj ff_ho3d ff_wide; do
  if [[ " $SPLIT " == *" $f "* ]]; then
    fl=; [[ $f == ff_cnf_fwd || $f == ff_wide ]] && fl="-mllvm -disable-machine-licm"
    (cd $SRC
     /opt/rocm/bin/hipcc $FLAGS $fl $EXTRA -S --cuda-device-only $f.hip -o $T/$f.s 2>/dev/null
     python3 $ROOT/tools/asm_lds_split.py $T/$f.s $T/${f}_split.s --stats
     $LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/${f}_split.s -o $T/${f}_dev.o
     $LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/${f}_dev.out $T/${f}_dev.o
     $LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
        -input=/dev/null -input=$T/${f}_dev.out -output=$T/$f.hipfb
     /opt/rocm/bin/hipcc $FLAGS $EXTRA --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/$f.hipfb -c $f.hip -o $T/$f.o) &
    OBJS="$OBJS $T/$f.o"
  else
    OBJS="$OBJS $SRC/$f.o"
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/fermiflow_amd/libfermiflow_hip_$NAME.so $OBJS -ldl
ls -la $ROOT/fermiflow_amd/libfermiflow_hip_$NAME.so
