#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s6; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bench_fallback.py "tests/test_gpu_graph.py::test_graphed_multiview_step_equals_eager_step" tests/test_gpu_solver.py -q -s -p no:cacheprovider > $O/tests.txt 2>&1; grep -E "SLIP gradient|FILIP gradient|passed|failed|FAILED" $O/tests.txt | cut -c1-250
