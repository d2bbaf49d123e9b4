mkdir -p gpurun_out/r02k
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02k
timeout 400 python -m pytest tests/test_gpu_1_kernels.py tests/test_gpu_2_decoder.py -q -x > $O/pytest_k.log 2>&1
timeout 300 python -m pytest tests/test_gpu_3_fullsize.py tests/test_gpu_4_model.py -q -x -k "parity or determinism or full_size" > $O/pytest_f.log 2>&1
timeout 200 python tools/kernel_probe.py --set fwd --iters 10 > $O/fwd.jsonl 2> $O/fwd.err
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest_k.log; tail -3 $O/pytest_f.log; cut -c1-110 $O/fwd.jsonl
python - <<PY
import json
for l in open("$O/bench.json"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"]); print(d["kernel_time_ms_per_step"]); print(d["hip_kernels_ms_per_step"])
PY
