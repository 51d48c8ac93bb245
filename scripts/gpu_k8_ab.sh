#!/bin/bash
# A/B of a filter_f16.hip compile-time variant on ONE box: the product library, then a scratch build with -D$1.
set -e
FLAG=${1:-FM_K8_R04_VISIT}
REPS=${2:-8}
python scripts/gpu_c5.py $REPS
cd fast-match_amd/csrc
cp ../libfastmatch_hip.so /tmp/lib_product.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -D$FLAG -c filter_f16.hip -o /tmp/f16_ab.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 rowreduce.o rounds.o dist_f32.o /tmp/f16_ab.o expand.o comm.o api_ctx.o api_match.o api_expand.o api_grid.o -ldl -o ../libfastmatch_hip.so
cd ../..
echo "---- with -D$FLAG"
python scripts/gpu_c5.py $REPS
cp /tmp/lib_product.so fast-match_amd/libfastmatch_hip.so
echo "---- product again"
python scripts/gpu_c5.py $REPS
