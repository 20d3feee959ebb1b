# kernel statistics of the default solver alone:  bash profiles/scripts/quick_trace_solve.sh cfg3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qts; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R:$R/tests
cd $R && python profiles/scripts/prof_lsmr_default.py "$@"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python $R/profiles/scripts/prof_lsmr_default.py "$@" > $O/run.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms over 4 solves" % (tot / 1e6))
for r in rows[:16]:
  print("%-58s calls %6s avg %8.2f us  %5.1f%%" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*.db" -delete
