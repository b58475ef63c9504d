cd $GRAFT_REPO_ROOT
for kv in "X=0" "GCRE_IE_BATCH=1" "GCRE_IE_BATCH=4" "GCRE_IE_BATCH=8" "GCRE_IE_WARM=512" "GCRE_IE_WARM=2048" "GCRE_IE_WARM=4096"; do
  echo "== $kv"; env $kv BENCH_ARGS="--method method2" STEPS=3 VARIANTS="-" tools/ab.sh
done
