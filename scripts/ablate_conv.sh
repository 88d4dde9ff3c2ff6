#!/bin/bash
# Build diagnostic variants of the library with one phase of conv_f16x3_kernel removed
# (HX_ABLATE bits, see amt_conv_f16x3.h) into amt-saga_amd/lib/ablate/, CPU side; then on the
# GPU box:   for v in 0 1 2 4 8 16; do AMT_LIB_PATH=amt-saga_amd/lib/ablate/libamt_$v.so python3 scripts/conv_microbench.py timing 512 2; done
# Results are wrong by construction; only the timings mean anything.
cd "$(dirname "$0")/.."
L=amt-saga_amd/lib; mkdir -p $L/ablate
python3 amt-saga_amd/build.py > /dev/null
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -Iinclude -Iamt-saga_amd/csrc \
     -DHX_ABLATE=$v -c amt-saga_amd/csrc/amt_rdcnn.hip -o $L/ablate/rdcnn_$v.o &
done
wait
for v in "$@"; do
  objs=$(ls $L/*.o | grep -v amt_rdcnn)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $L/ablate/libamt_$v.so $objs $L/ablate/rdcnn_$v.o
done
ls -la $L/ablate/*.so
