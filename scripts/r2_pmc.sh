# PMC passes on the batch workload (counters only + kernel-trace): bash scripts/r2_pmc.sh tag
export TMPDIR=/tmp
O=gpurun_out/$1
mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
( echo "# SQ pass 1"; bash scripts/pmc.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" A=1
  echo "# SQ pass 2"; bash scripts/pmc.sh "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" A=1
  echo "# SQ pass 3"; bash scripts/pmc.sh "SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES GRBM_GUI_ACTIVE" A=1
  echo "# FETCH_SIZE pass (KB; gfx950 reports 1/2 of wide coalesced reads: MI355X_MICROARCH.md HBM section)"; bash scripts/pmc.sh "FETCH_SIZE" A=1
  echo "# WRITE_SIZE / L2 pass"; bash scripts/pmc.sh "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" A=1 ) > $O/pmc.txt 2>&1
grep -v "^zv::\(embed\|add_\|rowdot\|bucket\|lr_\|stats\|norm_apply\)" $O/pmc.txt | cut -c1-330
