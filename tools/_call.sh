mkdir -p gpurun_out/r02u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02u
timeout 900 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 200 python tools/kernel_probe.py --set wgrad --iters 10 > $O/wgrad.jsonl 2> $O/wgrad.err
timeout 200 python tools/kernel_probe.py --set mid --iters 10 > $O/mid.jsonl 2> $O/mid.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --no-kernel-events > $O/bench_2proc_gloo.log 2>&1
tail -5 $O/pytest_gpu.log; tail -1 $O/smoke.log; cut -c1-100 $O/wgrad.jsonl; tail -2 $O/bench_2proc_gloo.log | cut -c1-400
python - <<PY
import json
print([ (json.loads(l)["case"], json.loads(l)["tflops"]) for l in open("$O/mid.jsonl") if l.startswith("{")])
for l in open("$O/bench_bf16.json"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline_lpg"]["frac"]); print(d["kernel_time_ms_per_step"]); print(d["hip_kernels_ms_per_step"], d["cpu_baseline"]["value"])
PY
