#!/bin/bash
# Cache-policy bits of the tile fills (global_load_lds aux operand): nt on every fill (aux2 = -DGEMM_AUX=2 build) against
# the default policy (base); pass further variant names as arguments.  Same process, interleaved rounds.
# build: tools/build_variants.sh base aux2:-DGEMM_AUX=2
set -e
echo "== config 1 shapes (cold weights, 12 copies)"
timeout -k 10 300 python3 tools/gemm_bench.py base aux2 $@ --rows 1408 --cold 12 2>&1 | grep -v amdgpu.ids
echo "== config 2 shapes"
timeout -k 10 300 python3 tools/gemm_bench.py base aux2 $@ 2>&1 | grep -v amdgpu.ids
echo "== Phi config 3 shapes"
timeout -k 10 300 python3 tools/gemm_bench.py base aux2 $@ --M 2560 --I 10240 --rows 27424 --gelu 1 --rounds 5 2>&1 | grep -v amdgpu.ids
