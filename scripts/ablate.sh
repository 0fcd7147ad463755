# phase ablation of the conv kernels (ZV_DBG bit 1: no staging, 2: no MFMA loop, 4: no epilogue, 8: no L2 warm-up,
# 16: every MFMA body re-reads the same weights); results are wrong numerically, only the durations matter.
# usage: DBGS="0 5 3" bash scripts/ablate.sh [grep pattern] [env assignments...]
export TMPDIR=/tmp
PAT=${1:-"20480,2,3"}
shift
for v in ${DBGS:-0 5 3 6 7}; do echo "== ZV_DBG=$v $@"; bash scripts/prof.sh ZV_DBG=$v "$@" | grep -E "$PAT" | awk '{print $1, $2, $3, $(NF-1)}'; done
