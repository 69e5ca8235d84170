#!/bin/bash
# Dev tool (GPU box): rocprofv3 passes of the SCALED config (BASELINE configs[4]), then its bench line with
# roofline.traffic measured from those passes:
#   tools/profile_scaled.sh NAME -> gpurun_out/prof_NAME (kernel trace + stats), gpurun_out/pmc_NAME/{fetch,write},
#                                   gpurun_out/bench_NAME.json
# One step is ~14,000 launches: the passes run 1 warm-up + 1 step; PMC counters one per pass, with --kernel-trace only.
name=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- python bench.py --config scaled --steps 1 --warmup 1 > gpurun_out/prof_$name.log 2>&1 || { tail -5 gpurun_out/prof_$name.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_$name/fetch -- python bench.py --config scaled --steps 1 --warmup 0 > gpurun_out/pmc_fetch_$name.log 2>&1 || { tail -5 gpurun_out/pmc_fetch_$name.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_$name/write -- python bench.py --config scaled --steps 1 --warmup 0 > gpurun_out/pmc_write_$name.log 2>&1 || { tail -5 gpurun_out/pmc_write_$name.log; exit 1; }
find gpurun_out/prof_$name gpurun_out/pmc_$name -name "*kernel_trace.csv" -size +20M -delete
python bench.py --config scaled --steps 2 --warmup 1 --pmc-dir gpurun_out/pmc_$name > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err || { tail -5 gpurun_out/bench_$name.err; exit 1; }
tail -c 1500 gpurun_out/bench_$name.json
