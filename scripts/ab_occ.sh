for v in occ4 occ3; do for m in exact fma; do
if [ $v = occ3 ]; then export BINF_LIB_OVERRIDE=$PWD/scripts/variants/occ3/libbinf_hip.so; else unset BINF_LIB_OVERRIDE; fi
python bench.py --steps 256 --warmup 64 --fuse 16 --mode $m --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('$v fuse=16 $m us/transition=%.2f steps/s=%.3e frac=%.3f'%(r['roofline']['avg_transition_us'], r['value'], r['roofline']['frac']))"
done; done
