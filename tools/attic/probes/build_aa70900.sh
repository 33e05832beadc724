#!/bin/bash
# Development aid: rebuild round 1's FAILING fused-upsampling halo kernel (commit aa70900) against today's other objects, so that the
# probes in this directory can run it:  APE_HIP_LIB=$PWD/autoposeestimation_amd/libape_hip_aa70900.so python tools/probes/ups_old_identity.py
# (DESIGN.md "The round-1 large-grid fault" has what they showed.)  Needs the git history; nothing here is loaded by the product path.
set -e
cd "$(dirname "$0")/../.."
T=$(mktemp -d)
mkdir -p $T/csrc $T/include
cp autoposeestimation_amd/csrc/*.h $T/csrc/
cp include/ape_hip.h $T/include/
git show aa70900:autoposeestimation_amd/csrc/conv3x3_halo.hip > $T/csrc/conv3x3_halo.hip
git show aa70900:autoposeestimation_amd/csrc/seg_head.h > $T/csrc/seg_head.h
git show aa70900:autoposeestimation_amd/csrc/common.h | sed 's#"../../include/ape_hip.h"#"../include/ape_hip.h"#' > $T/csrc/common.h
make -C autoposeestimation_amd/csrc -j4
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -c $T/csrc/conv3x3_halo.hip -o $T/halo_old.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o autoposeestimation_amd/libape_hip_aa70900.so \
    $(ls autoposeestimation_amd/csrc/build/*.o | grep -v conv3x3_halo) $T/halo_old.o
echo built autoposeestimation_amd/libape_hip_aa70900.so
