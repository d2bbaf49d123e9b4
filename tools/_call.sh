mkdir -p gpurun_out/r02e
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02e
for v in x y s; do
  BTS_CONV_BIG=$v timeout 300 python -m pytest tests/test_gpu_1_kernels.py -q -x -k "conv_fwd_dgrad_wgrad or conv_epilogues" > $O/pytest_conv_$v.log 2>&1
done
for v in a x y; do
  BTS_CONV_BIG=$v timeout 200 python tools/kernel_probe.py --set mid --iters 10 > $O/mid_$v.jsonl 2> $O/mid_$v.err
done
BTS_CONV_BIG=x BTS_PARITY_DUMP=$O/parity_x timeout 300 python -m pytest tests/test_gpu_3_fullsize.py -q -k "parity and bf16" > $O/pytest_parity_x.log 2>&1
BTS_CONV_BIG=y BTS_PARITY_DUMP=$O/parity_y timeout 300 python -m pytest tests/test_gpu_3_fullsize.py -q -k "parity and bf16" > $O/pytest_parity_y.log 2>&1
BTS_CONV_BIG=x timeout 300 python -m pytest tests/test_gpu_4_model.py -q -k "determinism" > $O/pytest_det_x.log 2>&1
BTS_CONV_BIG=y timeout 300 python -m pytest tests/test_gpu_4_model.py -q -k "determinism" > $O/pytest_det_y.log 2>&1
timeout 300 python -m pytest tests/test_gpu_1_kernels.py -q -x > $O/pytest_k1.log 2>&1
for v in x y s; do tail -2 $O/pytest_conv_$v.log; done
for v in a x y; do echo $v; python - <<PY
import json
print([ (json.loads(l)["case"], json.loads(l)["tflops"]) for l in open("$O/mid_$v.jsonl") if l.startswith("{")])
PY
done
tail -2 $O/pytest_parity_x.log; tail -2 $O/pytest_parity_y.log; tail -2 $O/pytest_det_x.log; tail -2 $O/pytest_det_y.log; tail -2 $O/pytest_k1.log
