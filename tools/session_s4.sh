mkdir -p gpurun_out/s4; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_block.py tests/test_bench_launch.py -x -q -m gpu -s > gpurun_out/s4/tests.log 2>&1; tail -6 gpurun_out/s4/tests.log
timeout 600 python -m pytest tests/test_gpu_resnet_intake_packed.py -x -q -m gpu -k "packed or pooled" > gpurun_out/s4/tests_packed.log 2>&1; tail -3 gpurun_out/s4/tests_packed.log
B="python bench.py --no-cpu-baseline --no-loss-delta --no-roofline --steps 30"
$B > gpurun_out/s4/resident_graph.json 2>gpurun_out/s4/resident_graph.err
$B --pipeline 1 > gpurun_out/s4/pipe_graph.json 2>gpurun_out/s4/pipe_graph.err
$B --pipeline 1 --graph 0 > gpurun_out/s4/pipe_eager.json 2>gpurun_out/s4/pipe_eager.err
for f in gpurun_out/s4/*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['host_ms_per_step'], d['config']['step_graph'], d['config'].get('graphs_captured'), d['loss'])" || tail -5 ${f%.json}.err; done
