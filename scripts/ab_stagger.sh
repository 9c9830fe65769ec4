for st in 0 1 2 4 8; do for m in exact fma; do
BINF_GAUSS_STAGGER=$st python bench.py --steps 512 --warmup 64 --fuse 64 --mode $m --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('stagger=$st fuse=64 $m us/transition=%.2f steps/s=%.3e frac=%.3f'%(r['roofline']['avg_transition_us'], r['value'], r['roofline']['frac']))"
done; done
