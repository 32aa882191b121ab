mkdir -p gpurun_out/s6; cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-loss-delta --no-roofline --steps 30"
run() { name=$1; shift; env "$@" $B > gpurun_out/s6/$name.json 2>gpurun_out/s6/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/s6/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])"; }
run base_a DH_X=0
run tail48_a DH_V4_TAIL=2 DH_V4_TAIL_NKT=48
run tail36_a DH_V4_TAIL=2 DH_V4_TAIL_NKT=36
run base_b DH_X=0
run tail48_b DH_V4_TAIL=2 DH_V4_TAIL_NKT=48
run dyn DH_V4_DYNAMIC=1
run onestream DH_TOWER_STREAMS=0
timeout 600 python -m pytest tests/test_gpu_golden_fullwidth.py -x -q -m gpu -k "defilip" > gpurun_out/s6/test_defilip.log 2>&1; tail -3 gpurun_out/s6/test_defilip.log
