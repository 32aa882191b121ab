#!/bin/bash
# round 6, session 17: the -m gpu suite on the final sources (as the driver runs it) + the default bench line with the driver's flags
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s17; mkdir -p $O
python -m declip_amd.build > /dev/null 2>&1
timeout 2400 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/gpu_suite.txt 2>&1; tail -8 $O/gpu_suite.txt | cut -c1-300
python bench.py --steps 20 --warmup 3 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; cut -c1-300 $O/bench_driver_flags.json
