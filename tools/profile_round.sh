# per-round evidence: bench JSON (default + twin-agent workload), rocprofv3 kernel stats and HBM traffic counters of the same command
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r02 v1
R=${1:-r02}; V=${2:-v1}; O=gpurun_out/$R$V; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
python bench.py --agents 2 --aug-copies 4 --steps 2 --warmup 1 --no-cpu-baseline --no-other-modes > $O/bench_twin.json 2> $O/bench_twin.err
rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python bench.py --no-cpu-baseline --no-other-modes --steps 3 --warmup 1 > $O/prof.log 2>&1
python tools/prof_summary.py $O/prof/p_results.db 70 > $O/step_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --no-cpu-baseline --no-other-modes --steps 2 --warmup 1 > $O/pmc_$c.log 2>&1
  python tools/pmc_summary.py $O/pmc_$c 30 > $O/pmc_${c}_summary.txt
  rm -rf $O/pmc_$c
done
rm -rf $O/prof
# SQ counters of the same command: effective clock, matrix-pipe duty, issue / wait split per kernel
for grp in "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/sq/$tag -- python bench.py --no-cpu-baseline --no-other-modes --steps 2 --warmup 1 > $O/sq_$tag.log 2>&1
done
python tools/pmc_clock.py $O/sq 24 > $O/sq_summary.txt 2>&1
rm -rf $O/sq $O/sq_*.log
# the voxelizer alone: kernel times and the HBM bytes one call really moves, incremental (persistent = 2: the training path) and
# stateless (persistent = 0: fresh grid, every cell written); 5 warm-up + 20 timed calls each -> divide the totals by 25
for pv in 0 2; do
  rocprofv3 --kernel-trace --stats -d $O/vprof$pv -o p -- python tools/bench_voxel.py --persistent $pv --iters 20 > $O/voxel_p$pv.json 2>/dev/null
  python tools/prof_summary.py $O/vprof$pv/p_results.db 12 > $O/voxel_p${pv}_kernel_stats.txt
  rm -rf $O/vprof$pv
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/vpmc -- python tools/bench_voxel.py --persistent $pv --iters 20 > /dev/null 2>&1
    python tools/pmc_summary.py $O/vpmc 12 > $O/voxel_p${pv}_pmc_${c}.txt
    rm -rf $O/vpmc
  done
done
head -14 $O/step_kernel_stats.txt | cut -c1-150
head -8 $O/pmc_FETCH_SIZE_summary.txt | cut -c1-150
python -c "
import json
for f in ('bench.json', 'bench_twin.json'):
    d = json.load(open('$O/' + f)); print(f, d['value'], d['ms_per_step'], d.get('samples_per_s'), d['roofline']['kernel'], round(d['roofline']['frac'], 3), d['rooflines_other'].get('voxel_scatter', {}).get('frac'))
"
