#!/bin/bash
# K8 ablation builds on ONE box (results of the ablated builds are wrong; only their times mean anything):
# product / no hand-over barrier / no exact path / neither / no fused rescoring / product again.
#   bash scripts/gpu_k8_ablate.sh > gpurun_out/k8_ablate.log
run() { python scripts/gpu_c5.py 2>&1 | grep -E "^(knn2|xcheck1)" | cut -c1-150; }
build() {
  ( cd fast-match_amd/csrc
    OBJS=""
    for u in rowreduce rounds dist_f32 filter_f16 knn_k expand comm api_ctx api_match api_expand api_grid; do
      if [ $u = filter_f16 ]; then
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $1 -c $u.hip -o /tmp/ab_$u.o
        OBJS="$OBJS /tmp/ab_$u.o"
      else OBJS="$OBJS $u.o"; fi
    done
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -ldl -o ../libfastmatch_hip.so )
}
cp fast-match_amd/libfastmatch_hip.so /tmp/lib_product.so
echo "---- product"; run
for f in "-DFM_ABLATE_K8_NORESCORE" "-DFM_ABL_RS_NOA" "-DFM_ABL_RS_NOB" "-DFM_ABL_RS_NOA -DFM_ABL_RS_NOB" "-DFM_ABLATE_F32_NOEXACT" "-DFM_ABLATE_K8_NORESCORE -DFM_ABLATE_K8_NOBARRIER"; do
  build "$f"; echo "---- $f"; run
done
cp /tmp/lib_product.so fast-match_amd/libfastmatch_hip.so
echo "---- product again"; run
