# produces the round's committed evidence: bench JSON, rocprofv3 kernel-trace summary, PMC passes, single-utterance chain
# usage (on the GPU box): bash scripts/round_profile.sh r02_v1      -> gpurun_out/<tag>/*  (copy into profiles/)
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
bash scripts/prof.sh A=1 > $OUT/${TAG}_kernel_trace_summary.txt
cp /tmp/zvprof/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null || true
( echo "# bench.py --steps 4 --no-graph --no-extras (batch of 32 x 1024 frames), counters averaged per dispatch and kernel configuration"
  echo "# SQ pass 1"; bash scripts/pmc.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" A=1
  echo "# SQ pass 2"; bash scripts/pmc.sh "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" A=1
  echo "# FETCH_SIZE pass (KB; gfx950 reports 1/2 of wide coalesced reads: MI355X_MICROARCH.md HBM section); ZV_TAIL_GROUPS=0: whole-batch launches, like the roofline leg of bench.py"; bash scripts/pmc.sh "FETCH_SIZE" ZV_TAIL_GROUPS=0
  echo "# WRITE_SIZE / L2 pass (ZV_TAIL_GROUPS=0)"; bash scripts/pmc.sh "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" ZV_TAIL_GROUPS=0 ) > $OUT/${TAG}_pmc.txt 2>&1
python scripts/chain.py > $OUT/${TAG}_full_chain.txt 2>&1
python scripts/traffic.py $OUT/${TAG}_pmc.txt $OUT/${TAG}_kernel_trace_summary.txt $TAG > $OUT/${TAG}_resblock_traffic.json
# bench.py quotes profiles/r04_resblock_traffic.json while its kernel_src_sha256 matches the source: re-run the bench with it in place
cp $OUT/${TAG}_resblock_traffic.json profiles/r04_resblock_traffic.json
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
bash scripts/r3_trace_gemm.sh A=1 > $OUT/${TAG}_one_pass_dispatches.txt 2>&1
cat $OUT/${TAG}_bench.json | head -c 1500; echo; cat $OUT/${TAG}_resblock_traffic.json
