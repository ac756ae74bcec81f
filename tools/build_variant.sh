#!/bin/bash
# A/B builds of one translation unit: tools/build_variant.sh <name> <unit> <-D flags...>
#   -> instascene_amd/libinstascene_hip_<name>.so = the base objects with <unit> recompiled with the flags.
# Select it at run time with ISR_LIB_PATH=<that file> (instascene_amd/_lib.py).
set -e
cd "$(dirname "$0")/../instascene_amd/csrc"
name=$1; unit=$2; shift 2
make -s -j5
obj=../../build/csrc
mkdir -p $obj/var_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function "$@" -c $unit.hip -o $obj/var_$name/$unit.o
others=$(ls $obj/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $others $obj/var_$name/$unit.o -o ../libinstascene_hip_$name.so
echo ../libinstascene_hip_$name.so
