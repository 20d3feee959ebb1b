# timeline of repeated solves (kernel + memory-copy trace): bash profiles/scripts/trace_solve.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ts; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R:$R/tests rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $O -o ts -- python $R/profiles/scripts/prof_solve_repeat.py cfg3 > $O/log.txt 2>&1
head -5 $O/log.txt; ls $O
