# default-bench artifacts for profiles/r1: the JSON line (no profiler) and the rocprofv3 --stats table of the same command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r1; mkdir -p $O
cd $R
timeout -k 10 500 python bench.py 2>/dev/null | tail -1 > $O/bench_r1_final.json || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py > $O/bench_prof.log 2>&1 || exit 1
tail -1 $O/bench_prof.log | cut -c1-300
