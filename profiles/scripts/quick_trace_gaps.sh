# durations of the four kernels of the bench step and the gaps between them (back-to-back launches):  bash profiles/scripts/quick_trace_gaps.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qth; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R:$R/tests
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o trace -- python $R/bench.py --no-scipy-mode --no-lsmr-mode --steps 200 --warmup 20 --no-cpu-baseline --no-solve > $O/bench.json 2> $O/bench.err
python - <<PY
import csv, glob, collections
f = glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mcba::", "")[:24] for r in rows]
prev_end = None
stats = collections.defaultdict(list)
for r, n in zip(rows, names):
  s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
  if prev_end is not None: stats[n].append((e - s, s - prev_end, prevn))
  prev_end, prevn = e, n
import statistics
for n in ("k_prep", "k_linearize<5, 0, 1, true", "k_assemble", "k_shared_final"):
  v = [x for x in stats[n] if x[1] < 20000]     # back-to-back launches only (gap < 20 us)
  d = sorted(x[0] for x in v); g = sorted(x[1] for x in v)
  if not d: continue
  q = lambda a, p: a[int(p * (len(a) - 1))]
  print("%-26s n=%5d  duration p10 %.2f p50 %.2f p90 %.2f us | gap to previous kernel p10 %.2f p50 %.2f p90 %.2f us | after %s" % (n, len(d), q(d,.1)/1e3, q(d,.5)/1e3, q(d,.9)/1e3, q(g,.1)/1e3, q(g,.5)/1e3, q(g,.9)/1e3, collections.Counter(x[2] for x in v).most_common(2)))
PY
rm -rf $O/trace
