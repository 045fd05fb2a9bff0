cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for k in i32 pk16; do
  d=gpurun_out/mb_$k; mkdir -p $d
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $d -o out --output-format csv -- python tools/sw_microbench.py 250 $k 1 > $d/log 2>&1
  d=gpurun_out/mb2_$k; mkdir -p $d
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD -d $d -o out --output-format csv -- python tools/sw_microbench.py 250 $k 1 > $d/log 2>&1
done
