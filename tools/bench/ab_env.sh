# A/B of an environment switch on the forward pass time: bash tools/bench/ab_env.sh VAR   (alternates unset / VAR=1, three rounds)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in "" "1"; do
    env ${v:+$1=1} python3 bench.py --steps 300 --warmup 10 --no-conv --no-cpu-baseline --adjoint-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=${v:-unset}', round(d['value']), round(d['ms_per_step']*1e3,1))"
  done
done
