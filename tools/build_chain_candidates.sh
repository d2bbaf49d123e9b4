#!/bin/bash
# Candidate libraries for the A/B of the r6 changes to lpg_chain_bwd_kernel (tools/chain_probe.py through tools/run_with_candidate.py):
#   g<0|1>dx<0|1>: head gradient requested at the top of the tile iteration (1) or at its use (0); old dx values batched (1) or per block (0)
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
mkdir -p /tmp/bts_chain bts_amd/lib/cand
OBJS=$(ls bts_amd/lib/*.o | grep -v lpg_chain.o)
for g in 0 1; do for d in 0 1; do
  /opt/rocm/bin/hipcc $F -DBTS_CHAIN_PRE_G=$g -DBTS_CHAIN_BATCH_DX=$d -c bts_amd/csrc/lpg_chain.hip -o /tmp/bts_chain/lpg_chain_g${g}dx${d}.o &
done; done
/opt/rocm/bin/hipcc $F -DBTS_CHAIN_DIAG_NOATOMIC=1 -c bts_amd/csrc/lpg_chain.hip -o /tmp/bts_chain/lpg_chain_noatomic.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bts_amd/lib/cand/libbts_amd_chain_noatomic.so /tmp/bts_chain/lpg_chain_noatomic.o $OBJS   # timing only: wrong weight gradients
for g in 0 1; do for d in 0 1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bts_amd/lib/cand/libbts_amd_chain_g${g}dx${d}.so /tmp/bts_chain/lpg_chain_g${g}dx${d}.o $OBJS
  echo bts_amd/lib/cand/libbts_amd_chain_g${g}dx${d}.so
done; done
