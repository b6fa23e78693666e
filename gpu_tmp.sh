python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 30 --warmup 3 --swarms 256 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('QP/s %.3e ms/step %.3f kernel_ms %.3f p50 %.3f'%(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['p50_replan_ms_one_swarm_host_to_host']), d['config']['launch_order'])"
python profiles/stamps.py 2>&1 | tail -32
