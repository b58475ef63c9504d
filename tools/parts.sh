cd $GRAFT_REPO_ROOT
for f in "" "-DGCRE_IEQ_NOEXACT" "-DGCRE_IEQ_NOPATHS"; do
  GCRE_EXTRA_FLAGS="$f" python3 geneticscre_amd/build.py > /dev/null 2>&1 || exit 1
  echo "flags [$f]"; tools/ktrace.sh 2>&1 | grep "k_null_ie_q" 
done
GCRE_EXTRA_FLAGS="" python3 geneticscre_amd/build.py > /dev/null 2>&1
