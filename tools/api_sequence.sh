# the HIP calls of ONE steady-state step in host order -- kernel launches, event records, stream waits, with their streams:
#   tools/api_sequence.sh <outdir-under-gpurun_out> <bench args...>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --hip-runtime-trace --kernel-trace --output-format json -d $O/tr -- python3 $R/bench.py --no-cpu-baseline --no-side --steps 12 --warmup 6 "$@" > $O/run.log 2>&1
find $O -name "*.db" -delete
python3 $R/tools/api_sequence_parse.py $O ${STEP_INDEX:-12}
