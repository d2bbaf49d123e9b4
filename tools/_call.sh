mkdir -p gpurun_out/r02j
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02j
timeout 200 python -m pytest tests/test_gpu_1_kernels.py -q -x -k "chain" > $O/pytest_chain.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 > $O/stats.log 2>&1
P="--graph 0 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o b -- python $GRAFT_REPO_ROOT/bench.py $P > $O/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o b -- python $GRAFT_REPO_ROOT/bench.py $P > $O/pmc_w.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_sq_fwd -o m -- python $GRAFT_REPO_ROOT/tools/kernel_probe.py --set fwd --iters 2 > $O/pmc_sq_fwd.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_sq_wgrad -o m -- python $GRAFT_REPO_ROOT/tools/kernel_probe.py --set wgrad --iters 2 > $O/pmc_sq_wgrad.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc_inst_fwd -o m -- python $GRAFT_REPO_ROOT/tools/kernel_probe.py --set fwd --iters 2 > $O/pmc_inst_fwd.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py $O/pmc_f/b_counter_collection.csv $O/pmc_w/b_counter_collection.csv $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
find $O -name '*kernel_trace.csv' -size +3M -delete
tail -3 $O/pytest_chain.log; ls $O/stats; cat $O/pmc_traffic.log; du -sh $O
