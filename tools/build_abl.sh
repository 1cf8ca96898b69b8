#!/bin/bash
# ab/libmrdis_abl.so: the library with the F(4x4) kernels built -DWINO4_ABLATIONS (timing-only variants selected by option debug_mode, in-kernel stamps);
# cross-compiles without a GPU.  Used by tools/wino4_abl.py, wino4_stamps.py, wgrad4_stamps.py.
set -e
cd "$(dirname "$0")/../representation-disentanglement_amd/csrc"
make -j8 libmrdis_hip.so > /dev/null
mkdir -p ../../ab
# same code generation flags as the production Makefile: no SLP vectorisation for mrdis_wino4 / mrdis_wino4r only
for f in mrdis_wino4 mrdis_wino4r; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -Wno-unused-value -Wno-unused-variable -DWINO4_ABLATIONS -c -o ../../ab/${f}_abl.o $f.hip
done
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-value -Wno-unused-variable -DWINO4_ABLATIONS -c -o ../../ab/mrdis_wino4w_abl.o mrdis_wino4w.hip
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/libmrdis_abl.so $(ls *.o | grep -v "mrdis_wino4.o\|mrdis_wino4r.o\|mrdis_wino4w.o") ../../ab/mrdis_wino4_abl.o ../../ab/mrdis_wino4r_abl.o ../../ab/mrdis_wino4w_abl.o
ls -la ../../ab/libmrdis_abl.so
# ab/libmrdis_abl_bf16.so: the pipelined bf16 convolution built -DBCONV3_ABLATIONS (tools/bconv_abl.py)
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-value -Wno-unused-variable -DBCONV3_ABLATIONS -c -o ../../ab/mrdis_bf16p_abl.o mrdis_bf16p.hip
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-value -Wno-unused-variable -DBCONV4_ABLATIONS -c -o ../../ab/mrdis_bf16q_abl.o mrdis_bf16q.hip
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/libmrdis_abl_bf16.so $(ls *.o | grep -v "mrdis_bf16p.o\|mrdis_bf16q.o") ../../ab/mrdis_bf16p_abl.o ../../ab/mrdis_bf16q_abl.o
# ab/libmrdis_abl_s6t.so: the six-product tap kernel built -DS6T_STAMPS (in-kernel stamps; tools/s6conv_stamps.py)
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-value -Wno-unused-variable -DS6T_STAMPS -c -o ../../ab/mrdis_s6conv_abl.o mrdis_s6conv.hip
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/libmrdis_abl_s6t.so $(ls *.o | grep -v "mrdis_s6conv.o") ../../ab/mrdis_s6conv_abl.o
