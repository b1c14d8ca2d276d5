# diagnostic: rebuild conv_pp.o with the clock stamps, measure, restore the production object (call through gpurun)
set -e
O=desenet_amd/csrc/build; cp $O/conv_pp.o /tmp/conv_pp.prod.o; cp desenet_amd/libdesenet_hip.so /tmp/lib.prod.so
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DDSN_PP_STAMP -c desenet_amd/csrc/conv_pp.hip -o $O/conv_pp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o desenet_amd/libdesenet_hip.so $O/*.o
python tools/exp/pp_clock.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pp_clock${PP_CLOCK_1X1:+_1x1}.txt
cp /tmp/conv_pp.prod.o $O/conv_pp.o; cp /tmp/lib.prod.so desenet_amd/libdesenet_hip.so
