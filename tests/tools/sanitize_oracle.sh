#!/bin/bash
# Host-side sanitizer pass (CPU only; GPU ASan is not available on this pool): builds the
# oracle with AddressSanitizer + UBSan and runs the oracle known-answer tests against it.
set -e
cd "$(dirname "$0")/../.."
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fopenmp -ffp-contract=off -fPIC -shared \
    -o /tmp/liboracle_asan.so oracle/bfmatch_oracle.c -lm
cp oracle/liboracle.so /tmp/liboracle_backup.so
cp /tmp/liboracle_asan.so oracle/liboracle.so
trap 'cp /tmp/liboracle_backup.so oracle/liboracle.so' EXIT
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) \
    python -m pytest tests/test_oracle_kat.py -x -q
