# forward ms per pass for a few settings of the solve loop's feed policy (LRNDE_FEED="T,E,M": rem <= T ? rem + E : rem/2 + E + 1
# launches kept beyond the last report read, never fewer than M)
cd $GRAFT_REPO_ROOT
for f in ${FEEDS:-"3,1,2" "3,2,3" "4,2,3" "6,2,3" "3,3,4" "100,1,2" "100,2,3"}; do
  for rep in 1 2; do
    LRNDE_FEED=$f python3 bench.py --steps 300 --warmup 10 --no-conv --no-cpu-baseline --adjoint-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"
  done
done
