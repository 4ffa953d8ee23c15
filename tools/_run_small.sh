mkdir -p gpurun_out/r5
for c in tmall_default cfg2 taobao_default ccmr_default; do
  for f in 0 512; do
    timeout -k 10 200 python bench.py --config $c --steps 300 --warmup 20 --no-cpu-baseline --no-side --debug-flags $f > gpurun_out/r5/b_${c}_$f.log 2>&1
    grep '^{' gpurun_out/r5/b_${c}_$f.log | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$c flags $f', round(j['value']), round(j['ms_per_step'],4), {k:(round(v,4) if v else v) for k,v in j['stages_ms'].items()})"
  done
done
timeout -k 10 200 python tools/hosttime_flags.py tmall_default 0 512 2>&1 | grep -v amdgpu
timeout -k 10 300 bash tools/kernel_sequence.sh r5/seq_tmall --config tmall_default > gpurun_out/r5/seq_tmall.log 2>&1; tail -60 gpurun_out/r5/seq_tmall.log
