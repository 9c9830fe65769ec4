for mode in exact fma; do for L in 1 5 10 20 40 80; do
  python bench.py --steps 200 --warmup 20 --nsteps $L --mode $mode --no-cpu-baseline | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('$mode L=$L us/launch=%.2f  steps/s=%.3e frac=%.3f'%(r['roofline']['avg_transition_us'], r['value'], r['roofline']['frac']))"
done; done
