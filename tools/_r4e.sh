python -m pytest tests/test_model_gpu.py -m gpu -x -q -s -k "step_cell or bf16 or fp32_parity or micro" > gpurun_out/r4e_tests.log 2>&1; echo "tests rc=$?"; grep -n "step cell\|bf16 3x1024\|bf16 step\|passed\|failed" gpurun_out/r4e_tests.log | tail -8
timeout -k 10 280 python bench.py --config scaled --steps 2 --warmup 1 > gpurun_out/r4e_scaled_new.json 2> gpurun_out/r4e_scaled_new.err; echo "scaled new rc=$?"
DEEPJ_STEP_EPILOGUE=0 timeout -k 10 280 python bench.py --config scaled --steps 2 --warmup 1 > gpurun_out/r4e_scaled_old.json 2> gpurun_out/r4e_scaled_old.err; echo "scaled old rc=$?"
python - <<'PY'
import json
for n in ("new", "old"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r4e_scaled_{n}.json") if l.startswith("{")][-1])
        print(n, d["ms_per_step"], d["roofline"]["whole_step_mfma_frac"], d["final_loss"], {k: v for k, v in d["kernel_ms_per_step"].items() if v > 20}, d["kernel_tflops"])
    except Exception as e:
        print(n, "failed", e); print(open(f"gpurun_out/r4e_scaled_{n}.err").read()[-800:])
PY
