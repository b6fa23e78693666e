export MASTER_ADDR=127.0.0.1 MASTER_PORT=29733 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
for g in 4 2 1; do
for f in "--force-dist"; do
python3 bench.py $f --no-extras --no-cpu-baseline --headline-only --groups $g 2>/dev/null | python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$f groups', d['config']['swarm_groups'], 'ms/step', round(d['ms_per_step'],4), 'host queue ms', round(d['config']['host_ms_to_queue_a_step'],4))
"
done
done
