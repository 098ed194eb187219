# Round-3 artifacts for profiles/r3 (run through gpurun from the repo root: `bash tools/bench/refresh_r3_profile.sh`):
#   1. the default bench line, no profiler                              -> bench_r3.json
#   2. rocprofv3 --kernel-trace --stats of the SAME command             -> bench_kernel_stats.csv, bench_kernel_trace_summary.txt
#   3. separate rocprofv3 --pmc passes (one counter set per pass, no other trace domain) over the MLP part of the bench
#      -> pmc/*.csv, pmc_summary.txt (FETCH/WRITE/L2 hit/MFMA busy + TA wave-loads, VMEM instructions, wait cycles)
#   4. conv field: --stats of the f-eval / VJP kernels (fp32 and bf16) and FETCH_SIZE / WRITE_SIZE passes -> conv/
#   5. traffic.json: what bench.py's roofline.traffic reads (MLP step kernel + the conv f-evals)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3prof; mkdir -p $O/pmc $O/conv
cd $R
timeout -k 10 500 python bench.py 2>/dev/null | tail -1 > $O/bench_r3.json || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --no-cpu-baseline > $O/bench_prof.log 2>&1 || exit 1
tail -1 $O/bench_prof.log | cut -c1-200
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE" \
            "TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_COALESCED_READ_CYCLES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAIT_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/pmc/$i -o p -- python3 $R/bench.py --no-conv --no-cpu-baseline --steps 5 --warmup 1 --adjoint-steps 0 --sustain-s 0 > $O/pmc/run$i.log 2>&1 || { echo "pmc pass $i ($ctrs) failed:"; tail -3 $O/pmc/run$i.log; }
done
cd $R
python3 profiles/pmc_summary.py $O/pmc "k_step_q<false" > $O/pmc_summary.txt
F=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 profiles/summarize.py $F > $O/bench_kernel_trace_summary.txt
cat $O/pmc_summary.txt $O/bench_kernel_trace_summary.txt
# ---- conv field ----
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/conv/prof_feval -o conv -- python3 $R/tools/bench/conv_bench.py > $O/conv/conv_bench.log 2>&1 || echo "conv_bench profile failed"
grep "us/f-eval" $O/conv/conv_bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/conv/prof_vjp -o bwd -- python3 $R/tools/bench/conv_bwd_bench.py > $O/conv/conv_vjp.log 2>&1 || echo "conv vjp profile failed"
for dt in f32 bf16; do
  j=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    j=$((j+1))
    rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/conv/pmc_${dt}_$j -o p -- python3 $R/tools/bench/conv_pmc_run.py $dt > $O/conv/pmc_${dt}_$j.log 2>&1 || echo "conv pmc $dt $ctrs failed"
  done
done
LRNDE_PMC_VJP=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/conv/pmc_vjp_1 -o p -- python3 $R/tools/bench/conv_pmc_run.py f32 > $O/conv/pmc_vjp_1.log 2>&1 || echo "conv vjp pmc failed"
LRNDE_PMC_VJP=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/conv/pmc_vjp_2 -o p -- python3 $R/tools/bench/conv_pmc_run.py f32 > $O/conv/pmc_vjp_2.log 2>&1 || echo "conv vjp pmc failed"
cd $R
python3 - <<PY
import csv, glob, os, re, json, shutil, collections
O = "$O"
# MLP: keep the merge small
f = glob.glob(O + "/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(O + "/bench_kernel_trace_step_rows.csv", "w", newline="") as g:
    w = csv.DictWriter(g, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp"]); w.writeheader()
    for r in rows:
        if "k_step" in r["Kernel_Name"]: w.writerow({k: r[k] for k in ("Kernel_Name", "Start_Timestamp", "End_Timestamp")})
s = glob.glob(O + "/prof/**/*kernel_stats.csv", recursive=True)
if s: os.replace(s[0], O + "/bench_kernel_stats.csv")
for c in glob.glob(O + "/pmc/**/*_counter_collection.csv", recursive=True):
    rr = [r for r in csv.DictReader(open(c)) if "k_step_q" in r["Kernel_Name"]]
    name = "_".join(sorted(set(r["Counter_Name"] for r in rr))) or "none"
    with open(O + f"/pmc/{name}_k_step_q.csv", "w", newline="") as g:
        w = csv.DictWriter(g, fieldnames=["Kernel_Name", "Counter_Name", "Counter_Value"]); w.writeheader()
        for r in rr: w.writerow({k: r[k] for k in ("Kernel_Name", "Counter_Name", "Counter_Value")})
vals = {}
for line in open(O + "/pmc_summary.txt"):
    m = re.match(r"(\w+)\s+dispatches\s+\d+\s+full-step mean\s+([0-9.]+)", line)
    if m: vals[m.group(1)] = float(m.group(2))
traffic = {"_how": "(2 x FETCH_SIZE + WRITE_SIZE) KiB from separate rocprofv3 --pmc passes (tools/bench/refresh_r3_profile.sh); FETCH_SIZE doubled "
                   "per the gfx950 correction of MI355X_MICROARCH.md (64 B counted per 128-B request on wide coalesced reads: an upper bound). "
                   "k_step_q_b512: per full-step launch of k_step_q<false,1>; conv_feval_*: summed over the five launches of one f-eval "
                   "(profiles/r3/conv/pmc_summary.txt)"}
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    traffic["k_step_q_b512"] = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
# conv: per-kernel means of each counter, per dtype; f-eval traffic = sum over its five launches
lines = []
def kernel_means(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(c)):
            m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
            if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
for dt in ("f32", "bf16"):
    tot = {}
    for j, ctr in ((1, "FETCH_SIZE"), (2, "WRITE_SIZE"), (3, None)):
        acc = kernel_means(O + f"/conv/pmc_{dt}_{j}")
        for k, v in sorted(acc.items()):
            lines.append(f"{dt:5s} {k:40s} " + "  ".join(f"{c}={sum(x)/len(x):.5g} (n={len(x)})" for c, x in v.items()))
            if ctr and ctr in v:
                per_feval = {"k_bn_finalize": 2}   # launches per f-eval of each kernel name: the finalize kernel runs twice
                nl = 2 if "bn_finalize" in k else 1
                tot[ctr] = tot.get(ctr, 0.0) + nl * sum(v[ctr]) / len(v[ctr])
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
        traffic[f"conv_feval_cifar_conv_{dt}_b256"] = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0
        lines.append(f"{dt:5s} one f-eval (conv1 + finalize + conv2 + finalize + conv3): FETCH_SIZE {tot['FETCH_SIZE']:.5g} KiB, WRITE_SIZE {tot['WRITE_SIZE']:.5g} KiB "
                     f"-> 2 x FETCH + WRITE = {traffic[f'conv_feval_cifar_conv_{dt}_b256']/1e6:.1f} MB")
for j in (1, 2):
    acc = kernel_means(O + f"/conv/pmc_vjp_{j}")
    for k, v in sorted(acc.items()):
        lines.append(f"vjp   {k:40s} " + "  ".join(f"{c}={sum(x)/len(x):.5g} (n={len(x)})" for c, x in v.items()))
open(O + "/conv/pmc_summary.txt", "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(O + "/traffic.json", "w"), indent=1)
for tag, pat in (("conv_feval_kernel_stats.csv", "/conv/prof_feval/**/*kernel_stats.csv"), ("conv_vjp_kernel_stats.csv", "/conv/prof_vjp/**/*kernel_stats.csv")):
    s = glob.glob(O + pat, recursive=True)
    if s: os.replace(s[0], O + "/conv/" + tag)
for d in glob.glob(O + "/pmc/[0-9]") + glob.glob(O + "/conv/pmc_*_[0-9]") + glob.glob(O + "/conv/prof_*") + [O + "/prof"]:
    shutil.rmtree(d, ignore_errors=True)
print(open(O + "/conv/pmc_summary.txt").read())
print(json.dumps(traffic, indent=1))
PY
