for i in 1 2 3; do timeout -k 10 200 python bench.py --workload kg --steps 2000 --warmup 100 --no-cpu-baseline --time-all 2>gpurun_out/kg_rep$i.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('kg', round(d['ms_per_step'],4), d['config'].get('rebuilds_in_timed_region'), d['config'].get('shortest_rebuild'))"; grep -v "^$" gpurun_out/kg_rep$i.err | head -8; done
