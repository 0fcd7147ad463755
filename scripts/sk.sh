# A/B of the split-K conv variants in one box: bash scripts/sk.sh   (SKV = "MT:LG:DBG ..." list)
export TMPDIR=/tmp
for v in ${SKV:-0:0:8 0:0:0 1:0:8 1:0:0 2:0:8 2:1:8 2:3:8}; do
  IFS=: read mt lgv dbg <<< "$v"
  echo "== ZV_SPLITK=$mt ZV_SPLITK_LG=$lgv ZV_DBG=$dbg"
  bash scripts/quick.sh ZV_SPLITK=$mt ZV_SPLITK_LG=$lgv ZV_DBG=$dbg | grep -v "^\[('voc_in"
  ZV_SPLITK=$mt ZV_SPLITK_LG=$lgv ZV_DBG=$dbg python scripts/chain.py 2>&1 | grep -E "hipGraph replay|batch 32" | head -3
done
