mkdir -p gpurun_out/r02h
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02h
timeout 300 python -m pytest tests/test_gpu_1_kernels.py -q -x -k "chain" > $O/pytest_chain.log 2>&1
timeout 300 python -m pytest tests/test_gpu_2_decoder.py -q > $O/pytest_dec.log 2>&1
BTS_PARITY_DUMP=$O/parity timeout 300 python -m pytest tests/test_gpu_3_fullsize.py -q -k "parity and bf16" > $O/pytest_parity.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest_chain.log; tail -15 $O/pytest_dec.log; tail -3 $O/pytest_parity.log
python - <<PY
import json
for l in open("$O/bench.json"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"]); print(d["roofline_lpg"]); print(d["kernel_time_ms_per_step"]); print(d["hip_kernels_ms_per_step"])
PY
