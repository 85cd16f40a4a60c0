# SQ counters of one attention-forward variant: bash tools/prof_flash2.sh <outdir> "<mode,waves,p[,shape]>" [lib]
O=$1; V=$2; LIB=${3:-}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ -n "$LIB" ] && export VOXACTB_HIP_LIB=$LIB
for grp in "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/x-$tag -- python tools/bench_flash2.py --one $V > $O/x-$tag.log 2>&1
done
python tools/pmc_clock.py $O 3 -v > $O/summary.txt 2>&1
rm -rf $O/x-*/ $O/x-*.log; cat $O/summary.txt
