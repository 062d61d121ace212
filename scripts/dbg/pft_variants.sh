#!/bin/bash
# times the tile force kernel with experiment builds of the library swapped in (scripts/dbg/libs/*.so)
cd $GRAFT_REPO_ROOT
cp hoomd-blue_amd/libhoomd_amd.so /tmp/keep.so
for v in stag1 stag2 full; do
  cp scripts/dbg/libs/$v.so hoomd-blue_amd/libhoomd_amd.so
  echo "== $v"
  timeout -k 10 200 python scripts/dbg/pft_time.py | tail -1 || exit 1
done
cp /tmp/keep.so hoomd-blue_amd/libhoomd_amd.so
