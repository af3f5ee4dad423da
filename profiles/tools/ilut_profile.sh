#!/bin/bash
# The profile build of the ILUT kernel (ilut_wp.hip with -DILUT_PROFILE) and its report on config C3:
#   bash profiles/tools/ilut_profile.sh build          (here: cross-compiles profiles/tools/lib_ilutprof.so)
#   bash profiles/tools/ilut_profile.sh run OUT.txt    (on the GPU box)
set -e
if [ "$1" = build ]; then
    cd /root/repo/ilupp_amd/csrc
    mkdir -p scratch
    F="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -Wall -Wno-unused-result"
    /opt/rocm/bin/hipcc $F -DILUT_PROFILE -c ilut_wp.hip -o scratch/ilut_prof.o
    OBJS=$(ls *.o | grep -v '^ilut_wp\.o$')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /root/repo/profiles/tools/lib_ilutprof.so $OBJS scratch/ilut_prof.o
    ls -la /root/repo/profiles/tools/lib_ilutprof.so
else
    ILUPP_HIP_LIBRARY=profiles/tools/lib_ilutprof.so python3 bench.py --steps 2 --warmup 1 --no-cpu --no-extra --config C3 2>&1 | grep "ilut profile" | tail -18 > "$2"
    cat "$2"
fi
