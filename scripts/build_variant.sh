#!/bin/bash
# Measurement build: the shipped objects + ONE translation unit recompiled with extra -D flags -> scripts/ubench/bin/libaudiolm_hip_<name>.so
# (git-ignored, travels with gpurun; loaded with ALM_LIB_PATH=...).  usage: scripts/build_variant.sh <name> <file.hip> [-DFLAG=V ...]
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/audiolm-pytorch_amd
python -c "import sys; sys.path.insert(0, '$pkg'); import build; build.build()" > /dev/null
mkdir -p $root/scripts/ubench/bin /tmp/alm_variant
obj=/tmp/alm_variant/${name}_${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c -o $obj $pkg/csrc/$src
others=$(ls $pkg/build/*.o | grep -v "/${src}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/scripts/ubench/bin/libaudiolm_hip_${name}.so $obj $others
echo built $root/scripts/ubench/bin/libaudiolm_hip_${name}.so
