#!/bin/bash
# build_variant.sh NAME FILE.hip "-DFLAG ..."  -> tools/_ablate/libsegsde_NAME.so: the shipped objects with FILE.hip recompiled with
# extra flags (kernel experiments; select with SEGSDE_LIB=tools/_ablate/libsegsde_NAME.so).  tools/_ablate is git-ignored but travels with gpurun.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FILE=$2; EXTRA=$3
OBJ=$ROOT/build/obj
OUT=$ROOT/tools/_ablate
mkdir -p $OUT $ROOT/build/var
EX=""
[ "$FILE" = "winograd_wgrad.hip" ] && EX="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-pass-failed $EX $EXTRA \
  -c $ROOT/improving_segmentation_with_selfsupervised_depth_amd/csrc/$FILE -o $ROOT/build/var/$NAME.o
OBJS=$(ls $OBJ/*.o | grep -v "/$FILE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $ROOT/build/var/$NAME.o -o $OUT/libsegsde_$NAME.so
echo $OUT/libsegsde_$NAME.so
