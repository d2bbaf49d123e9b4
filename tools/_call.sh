mkdir -p gpurun_out/r02s
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02s
BTS_CONV_WIDE=2 timeout 300 python -m pytest tests/test_gpu_1_kernels.py -q -x -k "conv_fwd_dgrad_wgrad or conv_epilogues" > $O/pytest_conv_w.log 2>&1
BTS_CONV_WIDE=1 timeout 200 python tools/kernel_probe.py --set mid --iters 10 > $O/mid_w1.jsonl 2> $O/mid_w1.err
timeout 300 python -m pytest tests/test_gpu_3_fullsize.py tests/test_gpu_4_model.py -q -x -k "parity or determinism or full_size" > $O/pytest_f_w.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_w.json 2> $O/bench_w.err
tail -2 $O/pytest_conv_w.log; tail -2 $O/pytest_f_w.log
python - <<PY
import json
print([ (json.loads(l)["case"], json.loads(l)["tflops"]) for l in open("$O/mid_w1.jsonl") if l.startswith("{")])
for l in open("$O/bench_w.json"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"]); print(d["kernel_time_ms_per_step"]); print(d["hip_kernels_ms_per_step"])
PY
