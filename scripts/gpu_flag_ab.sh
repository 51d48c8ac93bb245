#!/bin/bash
# A/B of a compile-time variant of ONE translation unit on ONE box: the product library, a scratch build of csrc/$1.hip with
# -D$2, the product again -- each running the command given after the first two arguments.
#   scripts/gpu_flag_ab.sh rowreduce FM_K1_SW python scripts/gpu_k12_ab.py nbuf=3
set -e
UNIT=$1; FLAG=$2; shift 2
"$@"
cd fast-match_amd/csrc
cp ../libfastmatch_hip.so /tmp/lib_product.so
OBJS=""
for u in rowreduce rounds dist_f32 filter_f16 knn_k expand comm api_ctx api_match api_expand api_grid; do
  if [ $u = $UNIT ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $( [ $u = dist_f32 ] && echo -fno-slp-vectorize ) -D$FLAG -c $u.hip -o /tmp/ab_$u.o
    OBJS="$OBJS /tmp/ab_$u.o"
  else
    OBJS="$OBJS $u.o"
  fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -ldl -o ../libfastmatch_hip.so
cd ../..
echo "---- with -D$FLAG"
"$@"
cp /tmp/lib_product.so fast-match_amd/libfastmatch_hip.so
echo "---- product again"
"$@"
