# usage (GPU box): bash tools/ab_run.sh "<lib names>" -- runs the GEMM micro-benchmark and the step's kernel table for each build, twice
for rep in 1 2; do for n in $1; do
  echo "== $n (rep $rep)"
  VOXACTB_HIP_LIB=gpurun_ab/lib_$n.so python tools/bench_gemm.py 2>&1 | grep bf16x3 | sed 's/staged.*dl+bfrag/dl+bfrag/' | cut -c1-120
  VOXACTB_HIP_LIB=gpurun_ab/lib_$n.so python bench.py --no-cpu-baseline --no-other-modes --steps 4 --warmup 2 --kernel-table 2>&1 >/dev/null | grep -i "polyphase\[\|gemm_fwd 32768x4096\|gemm_dgrad 32768x2048\|gemm_fwd 32768x1024" | cut -c1-100
done; done
