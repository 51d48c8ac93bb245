#!/bin/bash
# Scratch build on the GPU box (encoding shift switchable + in-kernel clock stamps), then scripts/gpu_k1_enc_ab.py
set -e
cd fast-match_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950"
/opt/rocm/bin/hipcc $F -DFM_ENC_AB -c api_ctx.hip -o /tmp/api_ctx_ab.o
/opt/rocm/bin/hipcc $F -DFM_CLOCK_STAMP -c rowreduce.hip -o /tmp/rr_ab.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/rr_ab.o rounds.o dist_f32.o filter_f16.o expand.o comm.o /tmp/api_ctx_ab.o api_match.o api_expand.o api_grid.o -ldl -o ../libfastmatch_hip.so
cd ../..
python scripts/gpu_k1_enc_ab.py
