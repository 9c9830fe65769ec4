for f in 1 4 16 64; do for m in exact fma; do
python bench.py --steps 256 --warmup 64 --fuse $f --mode $m --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('fuse=$f $m us/transition=%.2f steps/s=%.3e frac=%.3f acc=%.3f'%(r['roofline']['avg_transition_us'], r['value'], r['roofline']['frac'], r['acceptance_rate']))"
done; done
