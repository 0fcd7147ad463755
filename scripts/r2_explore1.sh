# round-2 exploration: the round-1 kernels at batch-like row counts, with the tile-shape overrides
export TMPDIR=/tmp
O=gpurun_out/r2e1
mkdir -p $O
echo "== default T=8192" > $O/log.txt
bash scripts/kprof.sh scripts/vocT.py 8192 >> $O/log.txt 2>&1
echo "== ZV_PAIR_MT=4 T=8192" >> $O/log.txt
ZV_PAIR_MT=4 bash scripts/kprof.sh scripts/vocT.py 8192 >> $O/log.txt 2>&1
echo "== ZV_NO_FUSE=1 ZV_CONV_MT=4 T=8192" >> $O/log.txt
ZV_NO_FUSE=1 ZV_CONV_MT=4 bash scripts/kprof.sh scripts/vocT.py 8192 >> $O/log.txt 2>&1
echo "== decoder T=4096" >> $O/log.txt
bash scripts/kprof.sh scripts/decT.py 4096 >> $O/log.txt 2>&1
echo "== decoder T=4096 ZV_CONV_MT=4" >> $O/log.txt
ZV_CONV_MT=4 bash scripts/kprof.sh scripts/decT.py 4096 >> $O/log.txt 2>&1
echo "== chain" >> $O/log.txt
python scripts/chain.py >> $O/log.txt 2>&1
echo done
