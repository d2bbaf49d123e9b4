mkdir -p gpurun_out/r02c
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02c
timeout 60 tools/probes/tr_probe > $O/tr_probe.log 2>&1
timeout 600 python -m pytest tests/test_gpu_1_kernels.py -q -x -k "conv_fwd_dgrad_wgrad" > $O/pytest_conv.log 2>&1
BTS_WGRAD_TR=1 timeout 300 python tools/kernel_probe.py --set wgrad --iters 10 > $O/wgrad_tr1.jsonl 2> $O/wgrad_tr1.err
BTS_WGRAD_TR=0 timeout 300 python tools/kernel_probe.py --set wgrad --iters 10 > $O/wgrad_tr0.jsonl 2> $O/wgrad_tr0.err
cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_lds -o w -- python $GRAFT_REPO_ROOT/tools/kernel_probe.py --set wgrad --iters 1 > $O/pmc_lds.log 2>&1
cd $GRAFT_REPO_ROOT
BTS_PARITY_DUMP=$O/parity timeout 900 python -m pytest tests/test_gpu_3_fullsize.py -q -k parity > $O/pytest_parity.log 2>&1
tail -3 $O/pytest_conv.log; cat $O/wgrad_tr1.jsonl | cut -c1-120; echo; cat $O/wgrad_tr0.jsonl | cut -c1-120; tail -4 $O/pytest_parity.log; head -8 $O/tr_probe.log
