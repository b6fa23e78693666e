# Developer tool: bench.py with N ranks on ONE GPU over gloo (bench.py's test hook): the multi-rank logic, not a measurement.
#   bash tests/tools/ranks_on_one_gpu.sh 8
N=${1:-8}
export DLSCQP_BENCH_GLOO_ONE_DEVICE=1 MASTER_ADDR=127.0.0.1
unset RANK LOCAL_RANK WORLD_SIZE
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29739 bench.py --gpus $N --swarms 8 --steps 6 --warmup 2 --no-cpu-baseline 2> gpurun_out/ranks_$N.err | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('n_gpus', d['n_gpus'], 'value', round(d['value']), 'groups', d['config']['swarm_groups'])
x=d['x2048_strong']; print('x2048: agents this rank', x['agents_this_rank'], 'interior', x['interior_agents_rank0'], 'halo', x['halo_agents_rank0'], 'solved', x['solved_fraction_rank0'])
e=d['empty60_sharded']; print('empty60:', {k:e[k] for k in ('agents','ranks','ms_per_step','qp_failsafes_last_step_all_ranks','mean_distance_flown_m')})
print('per_rank', {k:(round(v['min'],4),round(v['max'],4)) for k,v in d['per_rank_ms'].items()})
"
tail -3 gpurun_out/ranks_$N.err
