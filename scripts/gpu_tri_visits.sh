#!/bin/bash
# Counting build of the library (scratch copy on the GPU box), then scripts/gpu_tri_visits.py
set -e
cd fast-match_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DFM_COUNT_VISITS -c rowreduce.hip -o /tmp/rr_cnt.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/rr_cnt.o rounds.o dist_f32.o filter_f16.o expand.o comm.o api_ctx.o api_match.o api_expand.o api_grid.o -ldl -o ../libfastmatch_hip.so
cd ../..
python scripts/gpu_tri_visits.py "$@"
