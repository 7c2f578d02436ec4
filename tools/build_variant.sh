#!/bin/bash
# Developer A/B builds of the product library with extra -D flags: tools/build_variant.sh NAME [-DFLAG ...]
# -> build_exp/liblcs_NAME.so (git-ignored; travels to the GPU box; bench.py --lib loads it)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p build_exp
C=lte-cell-scanner_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-function "$@" \
  -o build_exp/liblcs_$NAME.so $C/pss_xcorr.hip $C/pss_xcorr_i8.hip $C/pss_xcorr_f16.hip $C/peak_search.hip $C/sss_foe.hip $C/tfg_mib.hip $C/tracker.hip $C/lcs_api.hip $C/lte_tables.cpp
