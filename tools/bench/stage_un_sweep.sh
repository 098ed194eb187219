cd $GRAFT_REPO_ROOT
for un in 4 7 13; do
  touch localregneuralde.jl_amd/csrc/lrnde_conv.hip
  make -C localregneuralde.jl_amd/csrc EXTRA=-DSTAGE_UN=$un > /dev/null 2>&1 || exit 1
  echo "== STAGE_UN=$un"
  timeout -k 10 200 python tools/bench/conv_bench.py 2>/dev/null | grep "f32 dbg=0 W=32 H=32 B=256\|f32 dbg=0 W=28"
  timeout -k 10 200 python tools/bench/conv_bwd_bench.py 2>/dev/null | tail -1
done
