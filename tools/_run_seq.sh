mkdir -p gpurun_out/r5
for c in cfg2 ccmr_default tmall_default; do
  timeout -k 10 300 bash tools/kernel_sequence.sh r5/seq_$c --config $c > gpurun_out/r5/seq_$c.log 2>&1
  echo "== $c"; cat gpurun_out/r5/seq_$c/sequence.txt
done
