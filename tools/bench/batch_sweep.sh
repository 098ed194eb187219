# step-kernel launch time and MFMA fraction over the batch size (profiles/r1/batch_sweep.txt)
cd $GRAFT_REPO_ROOT
echo "python bench.py --batch B --steps 5 --warmup 2 --adjoint-steps 0 --no-cpu-baseline --no-conv (MNIST-ODE MLP field, tol 1.4e-8, one MI355X):"
for B in 256 512 1024 2048 4096 8192; do
  timeout -k 10 300 python bench.py --batch $B --steps 5 --warmup 2 --adjoint-steps 0 --no-cpu-baseline --no-conv 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('B=$B  NFE/s %d  column-NFE/s %.3g  step kernel %.1f us  %.1f TFLOP/s  frac %.3f  (%s)' % (d['value'], d['value']*$B, r['us_per_launch'], r['achieved'], r['frac'], r['kernel'].split(' (')[0]))" || exit 1
done
