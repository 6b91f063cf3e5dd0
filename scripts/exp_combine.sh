for skip in 0 1; do for ch in 64 128 256 512; do
GNNAGG_DEBUG_SKIP_COMBINE=$skip TUNE_MODES=balanced TUNE_IDXMODE=1 TUNE_REMAP=2 TUNE_STREAM=0 TUNE_CHUNKS=$ch TUNE_ROUNDS=5 python scripts/tune_gcn.py 2>&1 | grep -E "plain|community" | awk -v s=$skip '$6!=0{print "skip="s, $0}'
done; done
