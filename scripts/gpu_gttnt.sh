#!/bin/bash
# A/B of the K2 genotype-load cache policy: rebuild libgpca.so on the box with -DGPCA_GTTX_AUX=2 (nt) and compare
set -e
export STEPS=10
bash scripts/gpu_ab.sh "1024 2048 i8 --storage int8" "1024 2048 i8 --storage 2bit"
touch genomic_pca_amd/csrc/gemm_i8.hip
make -C genomic_pca_amd/csrc -s CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fvisibility=hidden -DGPCA_GTTX_AUX=2" 2>&1 | grep -E "error" || true
echo "--- GPCA_GTTX_AUX=2"
bash scripts/gpu_ab.sh "1024 2048 i8 --storage int8" "1024 2048 i8 --storage 2bit"
