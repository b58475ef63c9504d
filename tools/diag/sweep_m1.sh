cd $GRAFT_REPO_ROOT
for kv in "X=0" "GCRE_IE_WARM=512" "GCRE_IE_WARM=256" "GCRE_IE_WARM_ITEMS=2" "GCRE_IE_WARM_ITEMS=8" "X=1"; do
  echo "== $kv"; env $kv STEPS=5 VARIANTS="-" tools/ab.sh
done
