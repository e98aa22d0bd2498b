#!/bin/bash
# Builds spiral_amd/lib/variants/libspmi_<name>.so from the tree as it is, replacing one kernel
# source by the file given: scripts/build_variant.sh <name> <source.hip> <replacement.hip> [hipcc flags]
# Use with SPMI_LIB_PATH=... to time kernel variants side by side on one box.
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; repl=$3; shift 3
mkdir -p spiral_amd/lib/variants /tmp/variant_$name
cp "$repl" /tmp/variant_$name/$src
cp spiral_amd/csrc/common.h /tmp/variant_$name/ 2>/dev/null || true
# common.h includes ../../include/spmi.h relative to csrc: compile in place with a temp name
tmp=spiral_amd/csrc/_variant_$name.hip
cp "$repl" $tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $tmp -o /tmp/variant_$name/obj.o
rm -f $tmp
objs=$(ls spiral_amd/lib/obj/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o spiral_amd/lib/variants/libspmi_$name.so $objs /tmp/variant_$name/obj.o
echo built spiral_amd/lib/variants/libspmi_$name.so
