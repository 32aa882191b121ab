#!/bin/bash
# tools/build_prev.sh [COMMIT]  ->  build/prev/libdeclip_hip.so built from the sources of COMMIT (default HEAD): the "before" arm of
# same-box A/B runs (tools/ab_bench.sh "prev:DECLIP_HIP_LIB=$PWD/build/prev/libdeclip_hip.so" "new:")
set -e
cd "$(dirname "$0")/.."
C=${1:-HEAD}
W=$(mktemp -d /tmp/dh_prev.XXXX)
git worktree add --detach "$W" "$C" > /dev/null 2>&1
(cd "$W" && python -m declip_amd.build > /dev/null 2>&1)
mkdir -p build/prev
cp "$W/declip_amd/libdeclip_hip.so" build/prev/libdeclip_hip.so
git worktree remove --force "$W"
git rev-parse "$C" > build/prev/COMMIT
ls -la build/prev/
