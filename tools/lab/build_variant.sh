#!/bin/bash
# A library that differs from the in-tree one in ONE translation unit compiled with extra flags:
#   tools/lab/build_variant.sh <name> <file.hip> <flags...>   ->  ab_libs/<name>.so   (then: tools/ab_lib2.sh ab_libs/a.so ab_libs/b.so)
set -e
NAME=$1; SRC=$2; shift 2
R=$(cd $(dirname $0)/../.. && pwd); C=$R/temporalalignnet_amd/csrc; O=$C/_obj
python -m temporalalignnet_amd.build > /dev/null
mkdir -p $R/ab_libs
/opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -fno-gpu-rdc -I$R/include -x hip -c $C/$SRC -o /tmp/variant_$NAME.o 2>/dev/null
OBJS=$(ls $O/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/variant_$NAME.o -o $R/ab_libs/$NAME.so
ls -la $R/ab_libs/$NAME.so
