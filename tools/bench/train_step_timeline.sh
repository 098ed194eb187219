# kernel timeline of the last training step of the MLP bench (forward with record on q0, companion stream, backward):
#   bash tools/bench/train_step_timeline.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/traintl_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python3 $R/bench.py --no-conv --no-cpu-baseline --steps 3 --warmup 1 --adjoint-steps 4 --sustain-s 0 > $O/run.log 2>&1 || { tail -3 $O/run.log; exit 1; }
F=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 $R/tools/bench/train_timeline.py $F > $O/timeline.txt
rm -rf $O/prof
head -60 $O/timeline.txt
