python -m pytest tests/test_model_gpu.py -m gpu -x -q -s -k "step_epilogue or bf16 or fp32_parity or micro" > gpurun_out/r4d_tests.log 2>&1; echo "tests rc=$?"; grep -n "step epilogue\|passed\|failed" gpurun_out/r4d_tests.log | tail -6
timeout -k 10 280 python bench.py --config scaled --steps 2 --warmup 1 > gpurun_out/r4d_scaled_new.json 2> gpurun_out/r4d_scaled_new.err; echo "scaled new rc=$?"
DEEPJ_STEP_EPILOGUE=0 timeout -k 10 280 python bench.py --config scaled --steps 2 --warmup 1 > gpurun_out/r4d_scaled_old.json 2> gpurun_out/r4d_scaled_old.err; echo "scaled old rc=$?"
python - <<'PY'
import json
for n in ("new", "old"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r4d_scaled_{n}.json") if l.startswith("{")][-1])
        print(n, d["ms_per_step"], d["roofline"]["whole_step_mfma_frac"], d["final_loss"], {k: v for k, v in d["kernel_ms_per_step"].items() if v > 20})
    except Exception as e:
        print(n, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args="--cpu-sample 0 --no-profile --gen-steps 0 --no-fp32 --scaled-steps 0"
export DEEPJ_LIB=$PWD/music-generator_amd/lib/libdeepj_hip.exh.so
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r04exh/fetch -- python bench.py --steps 2 --warmup 1 $args > gpurun_out/pmc_fetch_r04exh.log 2>&1; echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r04exh/write -- python bench.py --steps 2 --warmup 1 $args > gpurun_out/pmc_write_r04exh.log 2>&1; echo "pmc write rc=$?"
find gpurun_out/pmc_r04exh -name "*.csv" -size +20M -delete
