#!/bin/bash
# usage: bash tools/build_variant.sh <tag> <file.hip> "<extra flags>"  -- build adt_str_amd/libadt_exp_<tag>.so: the default objects with ONE
# translation unit recompiled under extra -D flags (A/B arms for ADT_LIB_PATH; run `make -C adt_str_amd/csrc` first)
set -e
TAG=$1; SRC=$2; FL=$3
C=$(cd "$(dirname "$0")/../adt_str_amd/csrc" && pwd)
B=${SRC%.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=fast $FL -c $C/$SRC -o /tmp/${B}_$TAG.o
OBJS=$(ls $C/*.o | grep -v "/$B.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/../libadt_exp_$TAG.so $OBJS /tmp/${B}_$TAG.o -lpthread
echo built libadt_exp_$TAG.so
