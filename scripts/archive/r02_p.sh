for S in 0 40 100; do for M in exact fma; do
BINF_GAUSS_STAGGER=$S python3 bench.py --fuse 1 --mode $M --steps 400 --warmup 100 --no-cpu-baseline --no-other-mode --no-extra --no-pmc 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('stagger $S $M fuse1: %.2f us/transition %.3e frac %.3f'%(r['avg_transition_us'],d['value'],r['frac']))"
done; done
