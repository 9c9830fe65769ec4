for rep in 1 2; do for v in block strided; do for f in 1 64; do
if [ $v = block ]; then unset BINF_LIB_OVERRIDE; else export BINF_LIB_OVERRIDE=$PWD/scripts/variants/$v/libbinf_hip.so; fi
python bench.py --fuse $f --mode exact --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('$v fuse=$f exact(recorded) us/transition=%.2f steps/s=%.3e'%(r['roofline']['avg_transition_us'], r['value']))"
done; done; done
