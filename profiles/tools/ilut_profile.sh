#!/bin/bash
# The profile build of the ILUT kernel (ilut_wp.hip with -DILUT_PROFILE) and its report on config C3:
#   bash profiles/tools/ilut_profile.sh build [FLAGS [NAME]]   (here: cross-compiles profiles/tools/lib_ilutprof.so, or lib_NAME.so with
#                                                               FLAGS, e.g. "-DILUT_PROFILE_SUB=1": sub-phase clocks in the two counter slots)
#   bash profiles/tools/ilut_profile.sh run OUT.txt [NAME]     (on the GPU box)
set -e
if [ "$1" = build ]; then
    cd /root/repo/ilupp_amd/csrc
    mkdir -p scratch
    F="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -Wall -Wno-unused-result"
    /opt/rocm/bin/hipcc $F -DILUT_PROFILE $2 -c ilut_wp.hip -o scratch/ilut_prof.o
    OBJS=$(ls *.o | grep -v '^ilut_wp\.o$')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /root/repo/profiles/tools/lib_${3:-ilutprof}.so $OBJS scratch/ilut_prof.o
    ls -la /root/repo/profiles/tools/lib_${3:-ilutprof}.so
else
    ILUPP_HIP_LIBRARY=profiles/tools/lib_${3:-ilutprof}.so python3 bench.py --steps 2 --warmup 1 --no-cpu --no-extra --config C3 2>&1 | grep "ilut profile" | tail -18 > "$2"
    cat "$2"
fi
