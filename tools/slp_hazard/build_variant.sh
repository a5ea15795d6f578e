#!/bin/bash
# Developer tool (root-causing the packed-fp32 non-determinism of tail_bf16.hip): builds tools/tail16_bench with the
# DEVICE code taken from an edited assembly file.
#   build_variant.sh <name> <sed-or-python filter reading dev asm on stdin, writing it to stdout> [extra hipcc device flags...]
# Steps: device asm (-S) -> filter -> assemble -> link (lld) -> bundle -> host compile with that fat binary embedded.
# The harness is compiled WITH packed fp32 arithmetic unless the caller passes the product's flags (-Xclang -target-feature -Xclang
# -packed-fp32-ops): reproducing the hazard is the point here, so the define that satisfies gem_internal.h's #error is always given.
set -euo pipefail
NAME=$1; FILTER=$2; shift 2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
LLVM=/opt/rocm/lib/llvm/bin
W=${SLP_WORK:-/tmp/slp}; mkdir -p $W
SRC=$ROOT/tools/tail16_bench.hip
KEY=$(echo "$* $(stat -c %Y $SRC $ROOT/globalegomocap_amd/csrc/*.h* | md5sum)" | md5sum | cut -c1-8)          # flags + source times
[ -f $W/dev_$KEY.s ] || /opt/rocm/bin/hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -DGEM_NO_PACKED_FP32 --cuda-device-only -S "$@" -o $W/dev_$KEY.s $SRC
bash -c "$FILTER" < $W/dev_$KEY.s > $W/$NAME.s
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/$NAME.s -o $W/$NAME.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/$NAME.out $W/$NAME.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/$NAME.out -output=$W/$NAME.hipfb
HOSTFLAGS=$(for f in "$@"; do case $f in -D*) echo $f;; esac; done)
/opt/rocm/bin/hipcc -w --offload-arch=gfx950 -O3 -std=c++17 --cuda-host-only -DGEM_NO_PACKED_FP32 $HOSTFLAGS -Xclang -fcuda-include-gpubinary -Xclang $W/$NAME.hipfb -c $SRC -o $W/$NAME.host.o
/opt/rocm/bin/hipcc -o $ROOT/tools/slp_hazard/t16_$NAME $W/$NAME.host.o
echo "built tools/slp_hazard/t16_$NAME: $(grep -c 'v_pk_[a-z]*_f32' $W/$NAME.s) packed fp32 instructions, $(grep -c 's_nop' $W/$NAME.s) s_nop"
