cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_v8 -o v8 -- python bench.py --no-cpu-baseline --no-other-modes --steps 3 --warmup 1 > gpurun_out/prof_v8.log 2>&1
python tools/prof_summary.py gpurun_out/prof_v8/v8_results.db 60 > gpurun_out/prof_v8_summary.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc8_$c -- python bench.py --no-cpu-baseline --no-other-modes --steps 2 --warmup 1 > gpurun_out/pmc8_$c.log 2>&1
  python tools/pmc_summary.py gpurun_out/pmc8_$c 30 > gpurun_out/pmc8_${c}_summary.txt
  rm -rf gpurun_out/pmc8_$c
done
head -12 gpurun_out/prof_v8_summary.txt | cut -c1-150
head -8 gpurun_out/pmc8_FETCH_SIZE_summary.txt | cut -c1-150
