# effective clock and issue/wait split of the conv-family kernels: rocprofv3 PMC passes over the two micro-benchmarks
# usage (GPU box): bash tools/pmc_clock.sh <outdir>
O=${1:-gpurun_out/pmc_clock}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for grp in "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  for b in halo wgrad_halo; do
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/$b-$tag -- python tools/bench_$b.py > $O/$b-$tag.log 2>&1
  done
done
python tools/pmc_clock.py $O 14 -v > $O/summary.txt 2>&1
rm -rf $O/*/ ; tail -60 $O/summary.txt
