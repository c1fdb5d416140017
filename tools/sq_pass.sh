# SQ-counter passes (rocprofv3 --pmc, with --kernel-trace only; 8 SQ slots per pass) over unpipelined frames of one workload
# (tools/probe.py): what bounds k_apply / k_apply_xlong / k_emit_lane — FETCH/WRITE alone cannot say.
#   bash tools/sq_pass.sh <workload> <frames> <out dir under gpurun_out>
W=${1:-C4-merged}; N=${2:-3}; TAG=${3:-sq_$W}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
P3="SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/p$i -o run -- python $R/tools/probe.py $W $N > $O/p$i.log 2>&1
  tail -2 $O/p$i.log | cut -c1-200
  i=$((i+1))
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/time -o run -- python $R/tools/probe.py $W $N > $O/time.log 2>&1
cd $R
python tools/sq_summarize.py $O "$W" | tee $O/summary.txt | cut -c1-220
find $O -name "*.csv" -size +2M -delete
