#!/bin/bash
# Candidate library with the r5 block -> (tile, split) mapping of the ring weight-gradient kernels (BTS_RING_XCD_SPLIT=0): the A side of
# the r6 A/B (bench.py --dump-launches through tools/run_with_candidate.py; FETCH_SIZE / WRITE_SIZE passes for the traffic)
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
mkdir -p /tmp/bts_ring bts_amd/lib/cand
OBJS=$(ls bts_amd/lib/*.o | grep -v conv_wgrad_tr.o)
/opt/rocm/bin/hipcc $F -DBTS_RING_XCD_SPLIT=0 -c bts_amd/csrc/conv_wgrad_tr.hip -o /tmp/bts_ring/conv_wgrad_tr_old.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bts_amd/lib/cand/libbts_amd_ring_oldmap.so /tmp/bts_ring/conv_wgrad_tr_old.o $OBJS
echo bts_amd/lib/cand/libbts_amd_ring_oldmap.so
