#!/bin/bash
# A/B of two builds of the library on ONE box: the product, another build (e.g. scripts/ablate/lib_r05.so = the previous
# round's sources built in a git worktree), the product again -- each running the command given after the first argument.
#   scripts/gpu_lib_ab.sh scripts/ablate/lib_r05.so python scripts/gpu_c5.py 5
OTHER=$1; shift
cp fast-match_amd/libfastmatch_hip.so /tmp/lib_product.so
echo "---- product"; "$@"
cp $OTHER fast-match_amd/libfastmatch_hip.so
echo "---- $OTHER"; "$@"
cp /tmp/lib_product.so fast-match_amd/libfastmatch_hip.so
echo "---- product again"; "$@"
