#!/bin/bash
# A/B of two builds of the library inside ONE gpurun call (boxes differ by 5-10 % in sustained store bandwidth, so
# only numbers of one call compare).  Builds the library of another git revision into tools/ab/ (git-ignored, travels
# to the GPU box like every other .so), to be loaded through BSX_NATIVE_LIB:
#   bash tools/ab_libs.sh <git-rev>            -> tools/ab/libbsuite_amd_prev.so
# then e.g.   gpurun -- 'for lib in tools/ab/libbsuite_amd_prev.so ""; do BSX_NATIVE_LIB=$lib python tools/lanes_sweep.py catch -- 2**17 2**20; done'
# Tuning knobs are read by the tuning build only: BSX_NATIVE_LIB=bsuite_amd/_lib/libbsuite_amd_tuning.so.
set -eu
rev=${1:-HEAD}
root=$(git rev-parse --show-toplevel)
rm -rf /tmp/bsx_ab && git worktree add -f /tmp/bsx_ab "$rev" >/dev/null
( cd /tmp/bsx_ab && python -m bsuite_amd.build | tail -1 )
mkdir -p "$root/tools/ab" && cp /tmp/bsx_ab/bsuite_amd/_lib/libbsuite_amd.so "$root/tools/ab/libbsuite_amd_prev.so"
git worktree remove --force /tmp/bsx_ab
ls -la "$root/tools/ab"
