# A/B of two builds on ONE GPU box: tools/ab_build.sh <name>  copies the current in-tree library to gpurun_ab/lib_<name>.so;
# run either with VOXACTB_HIP_LIB=gpurun_ab/lib_<name>.so (voxactb_amd/_lib.py).  gpurun_ab/ is git-ignored but travels.
mkdir -p gpurun_ab && python __graft_entry__.py build | tail -1 && cp voxactb_amd/csrc/libvoxactb_hip.so gpurun_ab/lib_$1.so && ls -la gpurun_ab/lib_$1.so
