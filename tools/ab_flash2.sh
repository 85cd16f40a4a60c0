# timing ablations of the round-4 attention forward: builds of flash2_fwd.hip with -DF2_ABLATE=<bits> (results are garbage, times are not)
# usage (build container): bash tools/ab_flash2.sh build "0 1 2 4 8 ..."   |   (GPU box): bash tools/ab_flash2.sh run "0 1 2 4 8 ..."
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p gpurun_ab
  for a in $1; do
    touch voxactb_amd/csrc/flash2_fwd.hip
    VXB_EXTRA_FLAGS="-DF2_ABLATE=$a" python -m voxactb_amd.csrc.build | tail -1
    cp voxactb_amd/csrc/libvoxactb_hip.so gpurun_ab/lib_f2a$a.so
  done
  touch voxactb_amd/csrc/flash2_fwd.hip; python -m voxactb_amd.csrc.build | tail -1
else
  for a in $1; do
    echo "== ablate $a"
    VOXACTB_HIP_LIB=gpurun_ab/lib_f2a$a.so python tools/bench_flash2.py --bench-only --self-only 2>&1 | grep flash2
  done
fi
