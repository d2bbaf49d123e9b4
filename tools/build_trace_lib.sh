#!/bin/bash
# Diagnostic library: libbts_amd.so with -DBTS_TRACE (s_memtime stamps in conv_igemm_dma's chunk loop), built beside the product
# binary as bts_amd/lib/libbts_amd_trace.so and used only through tools/run_with_candidate.py / tools/trace_igemm.py.
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
mkdir -p /tmp/bts_trace
/opt/rocm/bin/hipcc $F -DBTS_TRACE=1 -c bts_amd/csrc/conv_igemm.hip -o /tmp/bts_trace/conv_igemm.o
OBJS=$(ls bts_amd/lib/*.o | grep -v conv_igemm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bts_amd/lib/libbts_amd_trace.so /tmp/bts_trace/conv_igemm.o $OBJS
echo bts_amd/lib/libbts_amd_trace.so
