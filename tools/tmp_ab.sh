timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q 2>&1 | tail -2
for c in cfg3 cfg5_taobao; do for rep in 1 2 3; do for w in old new; do
  L=$GRAFT_REPO_ROOT/score_amd/lib/libscore_hip.so; [ $w = old ] && L=$GRAFT_REPO_ROOT/score_amd/lib/libscore_hip_old.so
  X="--steps 300"; [ $c = cfg5_taobao ] && X="--steps 10 --warmup 3 --batches 2"
  SCORE_HIP_LIB=$L python bench.py --no-side --no-cpu-baseline $X --config $c 2>/dev/null | grep '^{' | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$c $w', round(j['value']), round(j['ms_per_step'],4), round(j['stages_ms']['fwd_gru'],4))"
done; done; done
