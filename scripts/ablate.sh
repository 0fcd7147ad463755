export TMPDIR=/tmp
for v in 5 13 0 8; do echo "== ZV_DBG=$v"; ZV_DBG=$v ZEROVOX_AMD_LIB=$PWD/variants/lib_nob.so bash scripts/prof.sh A=1 | grep -E "pair" | awk '{print $1, $2, $NF, $(NF-1)}'; done
