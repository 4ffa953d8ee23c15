for w in 0 4 8 16 32; do for inl in "" 1; do
  [ "$w" = 0 ] && [ -n "$inl" ] && continue
  SCORE_ADAM_WINDOW=$w SCORE_ADAM_SWEEP_INLINE=$inl python bench.py --no-side --no-cpu-baseline --steps 100 2>/dev/null | grep '^{' | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); print('w=$w inline=$inl', round(j['value']), round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['stages_ms'].items()})"
done; done
