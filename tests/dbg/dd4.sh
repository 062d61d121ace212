#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for r in 0 1 2 3; do
  timeout -s USR1 -k 5 60 python -u tests/_dd_worker.py $r 4 29514 2 2 1 gpurun_out/dd4.npz 16 16 16 60 1 > gpurun_out/dd4_r$r.log 2>&1 &
done
wait
for r in 0 1 2 3; do echo "== rank $r"; grep -v "^\[W\|Gloo\|amdgpu.ids" gpurun_out/dd4_r$r.log | tail -n 14; done
