#!/bin/bash
# Build the committed HEAD of the HIP engine into build/alt_lib/ so that a GPU visit can compare it with the working tree
# (boxes differ by +-20% in clocks: only same-box A/B numbers are comparable).  usage: scripts/ab_prepare.sh [rev]
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/ab_base && git -C "$ROOT" worktree prune && git -C "$ROOT" worktree add -f /tmp/ab_base "$REV" >/dev/null 2>&1
make -C /tmp/ab_base/flac_amd/csrc -j8 >/dev/null 2>&1
mkdir -p "$ROOT/build/alt_lib" && cp /tmp/ab_base/flac_amd/lib/libflacgpu.so "$ROOT/build/alt_lib/libflacgpu.so"
git -C "$ROOT" worktree remove --force /tmp/ab_base
echo "alt lib = $REV"
