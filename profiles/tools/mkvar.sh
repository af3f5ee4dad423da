#!/bin/bash
# usage: profiles/tools/mkvar.sh NAME "-DFLAGS"  -> scratch/lib_NAME.so (rebuilds the sweep/factor kernels with the flags)
set -e
cd /root/repo/ilupp_amd/csrc
F="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -Wall -Wno-unused-result"
touch sptrsv.hip sptrsv_lm.hip ilu0.hip ilu0_lm.hip ilut.hip
make -j4 CXXFLAGS="$F $2" OUT=/root/repo/profiles/tools/lib_$1.so 2>&1 | grep -E " error|Error" || true
touch sptrsv.hip sptrsv_lm.hip ilu0.hip ilu0_lm.hip ilut.hip
ls -la /root/repo/profiles/tools/lib_$1.so
