mkdir -p gpurun_out/s5; cd $GRAFT_REPO_ROOT
timeout 300 python tools/prof_copies.py > gpurun_out/s5/prof_copies.txt 2>&1; tail -70 gpurun_out/s5/prof_copies.txt
timeout 900 python -m pytest tests/test_gpu_graph.py -x -q -m gpu -k "changing_caption" > gpurun_out/s5/test_graph.log 2>&1; tail -3 gpurun_out/s5/test_graph.log
timeout 900 python -m pytest tests/test_gpu_golden_fullwidth.py -x -q -m gpu -k "defilip" > gpurun_out/s5/test_defilip.log 2>&1; tail -5 gpurun_out/s5/test_defilip.log
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "slip_full" > gpurun_out/s5/test_slipw2.log 2>&1; tail -5 gpurun_out/s5/test_slipw2.log
