for m in exact fma; do for L in 1 10 20 40 80; do
python bench.py --steps 256 --warmup 64 --fuse 64 --nsteps $L --mode $m --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('persist fuse=64 $m L=$L us/transition=%.2f steps/s=%.3e'%(r['roofline']['avg_transition_us'], r['value']))"
done; done
