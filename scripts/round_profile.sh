# produces the round's committed evidence under profiles/: bench JSON, rocprofv3 kernel-trace summary, PMC passes
# usage (on the GPU box): bash scripts/round_profile.sh r01_v2
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
bash scripts/prof.sh A=1 > $OUT/${TAG}_kernel_trace_summary.txt
cp /tmp/zvprof/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null || true
( echo "# SQ pass"; bash scripts/pmc.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" A=1
  echo "# FETCH_SIZE pass (KB; gfx950 reports 1/2 of wide coalesced reads: MI355X_MICROARCH.md HBM section)"; bash scripts/pmc.sh "FETCH_SIZE" A=1
  echo "# WRITE_SIZE / L2 pass"; bash scripts/pmc.sh "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" A=1 ) > $OUT/${TAG}_pmc.txt
python scripts/chain.py > $OUT/${TAG}_full_chain.txt 2>&1
cat $OUT/${TAG}_bench.json
