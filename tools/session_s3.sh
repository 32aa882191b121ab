mkdir -p gpurun_out/s3; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_block.py -x -q -s > gpurun_out/s3/test_block.log 2>&1; tail -5 gpurun_out/s3/test_block.log
B="python bench.py --no-cpu-baseline --no-loss-delta --no-roofline --steps 20"
for mode in 0 1; do
  DH_BLOCK_NATIVE=$mode $B --graph 0 > gpurun_out/s3/eager_native$mode.json 2>gpurun_out/s3/eager_native$mode.err
  DH_BLOCK_NATIVE=$mode $B --graph 0 --pipeline 1 > gpurun_out/s3/pipe_native$mode.json 2>gpurun_out/s3/pipe_native$mode.err
done
$B > gpurun_out/s3/graph_default.json 2>gpurun_out/s3/graph_default.err
for f in gpurun_out/s3/*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['host_ms_per_step'], d['config']['step_graph'], d['config'].get('native_blocks'))"; done
