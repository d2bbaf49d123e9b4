#!/bin/bash
# Diagnostic libraries: libbts_amd.so with -DBTS_TRACE (s_memtime stamps in conv_igemm_dma's chunk loop), built beside the product
# binary as bts_amd/lib/libbts_amd_trace.so, plus two ablations of the same loop (timing only, results are garbage):
#   libbts_amd_trace_nodma.so   no LDS-DMA at all (MFMAs + fragment reads on whatever LDS holds)
#   libbts_amd_trace_nomfma.so  staging + fragment reads, no MFMAs
# Used only through tools/trace_igemm.py (BTS_TRACE_LIB selects one).
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
mkdir -p /tmp/bts_trace
OBJS=$(ls bts_amd/lib/*.o | grep -v conv_igemm.o)
for v in "" nodma nomfma; do
  D="-DBTS_TRACE=1"; S=""
  [ "$v" = nodma ] && D="$D -DBTS_ABL_NODMA=1" && S="_nodma"
  [ "$v" = nomfma ] && D="$D -DBTS_ABL_NOMFMA=1" && S="_nomfma"
  /opt/rocm/bin/hipcc $F $D -c bts_amd/csrc/conv_igemm.hip -o /tmp/bts_trace/conv_igemm$S.o &
done
wait
for S in "" _nodma _nomfma; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bts_amd/lib/libbts_amd_trace$S.so /tmp/bts_trace/conv_igemm$S.o $OBJS
  echo bts_amd/lib/libbts_amd_trace$S.so
done
