# Round-4 profile session (run on the GPU box through gpurun; outputs under gpurun_out/$1):
#   kernel stats of the steady-state bench at the driver's protocol, PMC traffic passes (FETCH_SIZE and WRITE_SIZE in SEPARATE
#   runs, no other trace domain beside --kernel-trace) of the bench workload and of the low-duplication gather probe.
# Condensed afterwards in the build container (git is there): tools/summarize_profile.py stats / pmc_round.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04p}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side > $O/stats.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side > $O/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side > $O/pmc_write.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/probe_fetch -- python3 $R/bench.py --gather-probe-only > $O/probe_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/probe_write -- python3 $R/bench.py --gather-probe-only > $O/probe_write.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/probe_stats -- python3 $R/bench.py --gather-probe-only > $O/probe_stats.log 2>&1
# keep what travels back small: the per-dispatch CSVs only
find $O -name "*.db" -delete
find $O -name "*_kernel_trace.csv" -size +20M -delete
du -sh $O
