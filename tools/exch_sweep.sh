cd $GRAFT_REPO_ROOT
export EXCH=1 CONFIG=sharded
echo "== tail 0 (old), E<=8"; GCRE_EXCHANGE_TAIL=0 python3 tools/rank_time.py 8 2>&1 | grep world
echo "== tail n/2, E<=8"; python3 tools/rank_time.py 8 2>&1 | grep world
echo "== tail n/2, E<=12 unit 31250"; GCRE_EXCHANGE_MAX=12 GCRE_EXCHANGE_UNIT=31250 python3 tools/rank_time.py 8 2>&1 | grep world
echo "== tail 8, E<=12 unit 31250"; GCRE_EXCHANGE_TAIL=8 GCRE_EXCHANGE_MAX=12 GCRE_EXCHANGE_UNIT=31250 python3 tools/rank_time.py 8 2>&1 | grep world
echo "== tail n/2, E<=16 unit 2000"; GCRE_EXCHANGE_MAX=16 GCRE_EXCHANGE_UNIT=2000 python3 tools/rank_time.py 8 2>&1 | grep world
