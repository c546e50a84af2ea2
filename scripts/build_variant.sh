#!/bin/bash
# A/B builds of libkicp.so for same-box comparisons (box-to-box noise is +-5 %): copies the sources (the working tree's, or a
# git revision's with REV=<rev>) into a scratch tree, applies the sed expressions given, builds, and leaves
# kiss-icp_amd/csrc/variants/libkicp_<name>.so (git-ignored, travels with gpurun).  Select with KICP_LIB=<path>.
#   scripts/build_variant.sh base                       # the working tree as it is
#   REV=HEAD scripts/build_variant.sh head              # the last commit
#   scripts/build_variant.sh fly8 's/kFly = 12/kFly = 8/'   # sed applied to every source file
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
w=/tmp/kicp_variants/$name
rm -rf "$w"; mkdir -p "$w/pkg/csrc" "$w/include"
if [ -n "$REV" ]; then
    git -C "$root" archive "$REV" kiss-icp_amd/csrc include | tar -x -C "$w/unpack.$$" 2>/dev/null || { mkdir -p "$w/unpack"; git -C "$root" archive "$REV" kiss-icp_amd/csrc include | tar -x -C "$w/unpack"; }
    cp "$w"/unpack/kiss-icp_amd/csrc/* "$w/pkg/csrc/"; cp "$w"/unpack/include/* "$w/include/"
else
    cp "$root"/kiss-icp_amd/csrc/*.hip "$root"/kiss-icp_amd/csrc/*.hpp "$root"/kiss-icp_amd/csrc/Makefile "$w/pkg/csrc/"; cp "$root"/include/* "$w/include/"
fi
for e in "$@"; do sed -i "$e" "$w"/pkg/csrc/*.hip "$w"/pkg/csrc/*.hpp; done
make -C "$w/pkg/csrc" -j8 >/dev/null
mkdir -p "$root/kiss-icp_amd/csrc/variants"
cp "$w/pkg/csrc/libkicp.so" "$root/kiss-icp_amd/csrc/variants/libkicp_$name.so"
echo "built kiss-icp_amd/csrc/variants/libkicp_$name.so"
