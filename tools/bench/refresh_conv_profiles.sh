# conv-field measurements for profiles/r1/conv: bench JSON lines per workload + the --stats table of conv_bench.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/conv_r1; mkdir -p $O
cd $R
for w in cifar_conv_bf16 cifar_conv_f32 cifar_conv_f32_split mnist_conv_f32 mnist_conv_f32_split; do
  timeout -k 10 400 python bench.py --workload $w 2>/dev/null | tail -1 > $O/bench_${w}.json || exit 1
  echo "$w done"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o conv -- python3 $R/tools/bench/conv_bench.py > $O/conv_bench.log 2>&1 || exit 1
grep "us/f-eval" $O/conv_bench.log
