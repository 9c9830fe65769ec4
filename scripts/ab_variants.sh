for v in base gs8 gs16; do for f in 1 64; do
if [ $v = base ]; then unset BINF_LIB_OVERRIDE; else export BINF_LIB_OVERRIDE=$PWD/scripts/variants/$v/libbinf_hip.so; fi
python bench.py --steps 1024 --warmup 128 --fuse $f --mode exact --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('$v fuse=$f exact(recorded) us/transition=%.2f steps/s=%.3e'%(r['roofline']['avg_transition_us'], r['value']))"
done; done
