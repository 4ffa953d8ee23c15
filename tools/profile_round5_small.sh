# kernel stats of the per-sample form at cfg-2 and the Tmall default shape (the last two lines of tools/profile_round5.sh alone)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05s}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg2 -- python3 $R/bench.py --config cfg2 --steps 400 --warmup 50 --no-cpu-baseline --no-side > $O/stats_cfg2.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_tmall_default -- python3 $R/bench.py --config tmall_default --steps 400 --warmup 50 --no-cpu-baseline --no-side > $O/stats_tmall_default.log 2>&1
find $O -name "*.db" -delete
find $O -name "*_kernel_trace.csv" -size +20M -delete
