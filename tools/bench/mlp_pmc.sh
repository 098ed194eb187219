# PMC passes over the default MLP bench (one rocprofv3 --pmc run per counter set, as the MI355X guide prescribes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mlp_pmc; mkdir -p $O
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/$i -o p -- python3 $R/bench.py --no-conv --no-cpu-baseline --steps 5 --warmup 1 --adjoint-steps 0 > $O/run$i.log 2>&1 || { tail -3 $O/run$i.log; exit 1; }
done
python3 $R/profiles/pmc_summary.py $O "k_step_q<false"
