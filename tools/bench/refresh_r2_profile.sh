# Round-2 artifacts for profiles/r2 (run through gpurun from the repo root):
#   1. the default bench line, no profiler                          -> bench_r2.json
#   2. rocprofv3 --kernel-trace --stats of the SAME command         -> bench_kernel_stats.csv, bench_kernel_trace.csv (step kernel rows)
#   3. separate rocprofv3 --pmc passes (one counter set per pass, no other trace domain) over the MLP part of the bench
#      -> pmc/*.csv, pmc_summary.txt, traffic.json (what bench.py's roofline.traffic reads)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2prof; mkdir -p $O/pmc
cd $R
timeout -k 10 400 python bench.py 2>/dev/null | tail -1 > $O/bench_r2.json || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py > $O/bench_prof.log 2>&1 || exit 1
tail -1 $O/bench_prof.log | cut -c1-200
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/pmc/$i -o p -- python3 $R/bench.py --no-conv --no-cpu-baseline --steps 5 --warmup 1 --adjoint-steps 0 > $O/pmc/run$i.log 2>&1 || { tail -3 $O/pmc/run$i.log; exit 1; }
done
cd $R
python3 profiles/pmc_summary.py $O/pmc "k_step_q<false" > $O/pmc_summary.txt
F=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 profiles/summarize.py $F > $O/bench_kernel_trace_summary.txt
cat $O/pmc_summary.txt $O/bench_kernel_trace_summary.txt
# keep the merge small: step-kernel rows of the per-dispatch trace only, counter CSVs reduced to the step kernel
python3 - <<PY
import csv, glob, os
O = "$O"
f = glob.glob(O + "/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
keep = [r for r in rows if "k_step" in r["Kernel_Name"]]
with open(O + "/bench_kernel_trace_step_rows.csv", "w", newline="") as g:
    w = csv.DictWriter(g, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp"]); w.writeheader()
    for r in keep: w.writerow({k: r[k] for k in ("Kernel_Name", "Start_Timestamp", "End_Timestamp")})
s = glob.glob(O + "/prof/**/*kernel_stats.csv", recursive=True)
if s: os.replace(s[0], O + "/bench_kernel_stats.csv")
for c in glob.glob(O + "/pmc/**/*_counter_collection.csv", recursive=True):
    rr = [r for r in csv.DictReader(open(c)) if "k_step_q" in r["Kernel_Name"]]
    name = "_".join(sorted(set(r["Counter_Name"] for r in rr))) or "none"
    with open(O + f"/pmc/{name}_k_step_q.csv", "w", newline="") as g:
        w = csv.DictWriter(g, fieldnames=["Kernel_Name", "Counter_Name", "Counter_Value"]); w.writeheader()
        for r in rr: w.writerow({k: r[k] for k in ("Kernel_Name", "Counter_Name", "Counter_Value")})
import shutil, json, re
vals = {}
for line in open(O + "/pmc_summary.txt"):
    m = re.match(r"(\w+)\s+dispatches\s+\d+\s+full-step mean\s+([0-9.]+)", line)
    if m: vals[m.group(1)] = float(m.group(2))
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    json.dump({"k_step_q_b512": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
               "_how": "profiles/r2/pmc_summary.txt: (2 x FETCH_SIZE + WRITE_SIZE) KiB per full-step launch of k_step_q<false,1> at B=512, "
                       "separate rocprofv3 --pmc passes (tools/bench/refresh_r2_profile.sh); FETCH_SIZE doubled per the gfx950 correction of "
                       "MI355X_MICROARCH.md (64 B counted per 128-B request on wide coalesced reads: an upper bound)"},
              open(O + "/traffic.json", "w"), indent=1)
for d in glob.glob(O + "/pmc/[0-9]"): shutil.rmtree(d)
shutil.rmtree(O + "/prof", ignore_errors=True)
PY
