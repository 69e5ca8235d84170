#!/bin/bash
# Dev tool (GPU box): the rocprofv3 passes a round commits under profiles/, then the bench line of the same build with
# roofline.traffic MEASURED from those passes (bench.py --pmc-dir):
#   tools/profile_round.sh NAME   -> gpurun_out/prof_NAME (kernel trace + stats), gpurun_out/pmc_NAME/{fetch,write},
#                                    gpurun_out/bench_NAME.json
# PMC counters are collected in runs of their own (with --kernel-trace only), one counter per pass.
name=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args="--cpu-sample 0 --no-profile --gen-steps 0 --no-fp32 --scaled-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- python bench.py --steps 5 --warmup 2 $args > gpurun_out/prof_$name.log 2>&1 || { tail -5 gpurun_out/prof_$name.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_$name/fetch -- python bench.py --steps 2 --warmup 1 $args > gpurun_out/pmc_fetch_$name.log 2>&1 || { tail -5 gpurun_out/pmc_fetch_$name.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_$name/write -- python bench.py --steps 2 --warmup 1 $args > gpurun_out/pmc_write_$name.log 2>&1 || { tail -5 gpurun_out/pmc_write_$name.log; exit 1; }
# keep only the small summaries (the traces are tens of MB)
find gpurun_out/prof_$name gpurun_out/pmc_$name -name "*.csv" -size +20M -delete
python bench.py --pmc-dir gpurun_out/pmc_$name "${@:2}" > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err || { tail -5 gpurun_out/bench_$name.err; exit 1; }
tail -c 3000 gpurun_out/bench_$name.json
