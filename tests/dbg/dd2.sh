#!/bin/bash
# two ranks on one GPU, output per rank into gpurun_out/
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1
timeout -k 5 120 python -u -X faulthandler tests/_dd_worker.py 0 1 29511 1 1 1 gpurun_out/ref.npz 16 16 16 60 0 > gpurun_out/dd_ref.log 2>&1
echo "single exit=$?"
for r in 0 1; do
  timeout -s USR1 -k 5 100 python -u tests/_dd_worker.py $r 2 29512 1 1 2 gpurun_out/dd.npz 16 16 16 60 0 > gpurun_out/dd_r$r.log 2>&1 &
done
wait
echo "dd done"
for f in gpurun_out/dd_ref.log gpurun_out/dd_r0.log gpurun_out/dd_r1.log; do tail -n 12 $f; done
python - <<'PY'
import numpy as np
a=np.load("gpurun_out/ref.npz"); b=np.load("gpurun_out/dd.npz")
L=16*(1/0.84)**(1/3)
print("e0",a["e0"],b["e0"]); print("e1",a["e1"],b["e1"])
print("dpos",np.abs(a["position"]+a["image"]*L-b["position"]-b["image"]*L).max(),"dvel",np.abs(a["velocity"]-b["velocity"]).max())
for k in ("N_local","n_ghosts","builds","migrated","dof"): print(k,a[k],b[k])
PY
