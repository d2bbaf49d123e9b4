#!/bin/bash
# Round-end measurement set on the binary in bts_amd/lib (run on the GPU box through gpurun; everything lands in gpurun_out/).
#   bash tools/final_protocol.sh <tag>
# pytest -m gpu + smoke, the bench lines of every BASELINE configuration, rocprofv3 kernel statistics of the bench command and the
# two PMC passes behind profiles/pmc_traffic.json (BTS_CONV_WIDE=0: rocprofv3 aborts a counter pass when conv_halo_wide's 160 KiB
# of dynamic LDS are dispatched, profiles/r03_pmc_fetch_abort_with_halo_wide.log).
T=${1:-final}
O=gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
md5sum bts_amd/lib/libbts_amd.so > $O/${T}_pytest_gpu.log
timeout 600 python -m pytest tests -m gpu -x -q >> $O/${T}_pytest_gpu.log 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/${T}_pytest_gpu.log 2>&1
tail -3 $O/${T}_pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 --dump-launches $O/${T}_launches.json > $O/${T}_bench_bf16.json 2> $O/${T}_bench_bf16.err
timeout 200 python bench.py --dtype f32 --no-cpu-baseline --lpg-op 0 > $O/${T}_bench_f32.json 2> /dev/null
timeout 200 python bench.py --height 416 --width 544 --batch 16 --dataset nyu --no-cpu-baseline --lpg-op 0 > $O/${T}_bench_c2.json 2> /dev/null
timeout 300 python bench.py --encoder resnext101_bts --dtype f32 --no-cpu-baseline --lpg-op 0 > $O/${T}_bench_c4.json 2> /dev/null
timeout 300 python bench.py --mode infer --height 704 --width 1216 --batch 32 > $O/${T}_bench_infer.json 2> /dev/null
for f in bf16 f32 c2 c4 infer; do cut -c1-160 $O/${T}_bench_$f.json; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -o b -- python bench.py --no-cpu-baseline --steps 10 --warmup 3 --parity 0 --lpg-op 0 > $O/${T}_bench_under_rocprof.json 2> $O/${T}_rocprof.err
cp $(find $O/${T}_prof -name '*kernel_stats.csv' | head -1) $O/${T}_bench_kernel_stats.csv 2> /dev/null
rm -rf $O/${T}_prof
for c in FETCH_SIZE WRITE_SIZE; do
  BTS_CONV_WIDE=0 timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${T}_pmc_$c -o b -- python bench.py --graph 0 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --parity 0 --lpg-op 0 > $O/${T}_pmc_$c.out 2> $O/${T}_pmc_$c.err
  echo "pmc $c rc=$?"
  cp $(find $O/${T}_pmc_$c -name '*counter_collection.csv' | head -1) $O/${T}_pmc_$c.csv 2> /dev/null
  rm -rf $O/${T}_pmc_$c
done
ls -la $O | grep ${T}_ | head -30
