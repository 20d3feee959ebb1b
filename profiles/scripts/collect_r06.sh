# Round-6 evidence, ALL of it from the commit that is checked out:  bash profiles/scripts/collect_r06.sh   (on the GPU box)
# Raw rocprofv3 output goes to gpurun_out/r6p (scratch); `python profiles/make_summary.py r06 gpurun_out/r6p` condenses it into the
# tracked profiles/r06_* files: kernel statistics of the bench (evaluation step) and of the LSMR iteration in its three forms, PMC
# averages of every kernel of both, the native solver's LM timeline, Workspace.calibrate under both solvers, the lsmr-mode table.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R:$R/tests
B="python $R/bench.py --no-scipy-mode --no-lsmr-mode"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- $B --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_traced.json 2> $O/bench_traced.err
# PMC: counters in their own passes, kernel trace only (no other trace domain)
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -o pmc_fetch -- $B --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-solve > /dev/null 2> $O/pmc1.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc -o pmc_write -- $B --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-solve > /dev/null 2> $O/pmc2.err
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc -o pmc_mfma -- $B --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-solve > /dev/null 2> $O/pmc3.err
# (`bash profiles/scripts/collect_r06.sh bench`: only the evaluation step -- kernel statistics, PMC, bench line)
if [ "$1" != "bench" ]; then
# the LSMR iteration (default solver): kernel trace of the three forms + PMC of the kernels
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lsmr_trace -o lsmr -- python $R/profiles/scripts/prof_lsmr_iter.py cfg3 > $O/lsmr_traced.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/lsmr_pmc_$C -o pmc -- python $R/profiles/scripts/prof_lsmr_iter.py cfg3 > $O/lsmr_pmc_$C.log 2>&1
done
timeout 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/lsmr_pmc_valu -o pmc -- python $R/profiles/scripts/prof_lsmr_iter.py cfg3 > $O/lsmr_pmc_valu.log 2>&1
for CFG in cfg3 cfg4; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/solve_$CFG -o solve -- python $R/profiles/scripts/prof_cfg.py $CFG > $O/solve_$CFG.log 2>&1
done
cd $R; timeout 120 python profiles/scripts/prof_lsmr_iter.py cfg3 cfg4 cfg5 cfg2 > $O/lsmr_iteration.log 2>&1; cut -c1-400 $O/lsmr_iteration.log
fi
cd $R
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.json
if [ "$1" != "bench" ]; then
timeout 200 python bench.py --config cfg4 --no-cpu-baseline --no-scipy-mode > $O/bench_cfg4.json 2> $O/bench_cfg4.err
for CFG in cfg3 cfg4 cfg2; do
  MCBA_TIMING=1 timeout 100 python profiles/scripts/prof_workspace.py $CFG --solver native > $O/workspace_$CFG.log 2>&1; grep "calibrate ms" $O/workspace_$CFG.log
  timeout 100 python profiles/scripts/prof_workspace.py $CFG --solver lsmr > $O/workspace_lsmr_$CFG.log 2>&1; grep "calibrate ms" $O/workspace_lsmr_$CFG.log
done
# round 6: k_linearize over the compacted observation tables against the masks form, and the persistent grid
(python profiles/scripts/prof_lin_compact.py; echo "masks form (MCBA_LIN_COMPACT=0)"; TEST_MASKS=1 python profiles/scripts/prof_lin_compact.py; echo "persistent grid"; python profiles/scripts/prof_lin_grid.py) > $O/lin_compact.txt 2>&1; cp $O/lin_compact.txt profiles/r06_lin_compact.txt; cat $O/lin_compact.txt
# round 6: the sign of the default solver's offset from the reference's end point (call level, bisection, summation orders)
timeout 900 python profiles/scripts/prof_lsmr_sign.py AB > $O/lsmr_sign.json 2> $O/lsmr_sign.log; tail -8 $O/lsmr_sign.log
timeout 600 python profiles/scripts/prof_lsmr_bisect.py > $O/lsmr_bisect.json 2> $O/lsmr_bisect.log; tail -16 $O/lsmr_bisect.log
timeout 300 python -m pytest tests/test_gpu_lsmr.py -q -s -k "call_matches or call_sequence or converged_optimum or exact_product" > $O/lsmr_call_parity.log 2>&1; grep -E "istop|device - scipy|tight optimum|passed|failed" $O/lsmr_call_parity.log | cut -c1-260 > $O/lsmr_call_parity.txt
LSMR_NO_SCIPY=1 timeout 300 python profiles/scripts/prof_lsmr.py > $O/lsmr_mode.md 2> $O/lsmr_mode.err; tail -5 $O/lsmr_mode.md
timeout 500 python profiles/scripts/prof_parity_table.py > $O/parity_table.md 2> $O/parity_table.err; cp gpurun_out/parity_table.json $O/ 2>/dev/null
fi
# condense ON THE BOX (gpurun copies back at most 64 MiB: the raw kernel traces are ~30 MB each) and keep only the summaries
python profiles/make_summary.py r06 $O > $O/make_summary.log 2>&1
cp $O/lsmr_sign.json profiles/r06_lsmr_sign.json; cp $O/lsmr_bisect.json profiles/r06_lsmr_bisect.json; cp $O/lsmr_call_parity.txt profiles/r06_lsmr_call_parity.txt
mkdir -p $R/gpurun_out/r06 && cp profiles/r06_* profiles/hbm_traffic.json profiles/parity_table.md profiles/parity_table.json $R/gpurun_out/r06/ 2>/dev/null
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*_counter_collection.csv" -delete; find $O -name "*.db" -delete
du -sh $R/gpurun_out; ls $R/gpurun_out/r06
