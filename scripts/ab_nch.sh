# A/B of the chain-slots-per-wave variants of the fused Gaussian kernel
for nch in 1 2; do for mode in exact fma; do for L in 1 20; do
  BINF_GAUSS_NCH=$nch python bench.py --steps 300 --warmup 30 --nsteps $L --mode $mode --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('NCH=$nch $mode L=$L us/launch=%.2f  steps/s=%.3e frac=%.3f'%(r['roofline']['avg_transition_us'], r['value'], r['roofline']['frac']))"
done; done; done
