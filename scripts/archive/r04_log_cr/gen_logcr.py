import mpmath as mp
mp.mp.prec = 400
def dd(x):
    hi = float(x); lo = float(x - mp.mpf(hi)); return hi, lo
NT = 22
lines = []
for n in range(NT):
    h, l = dd(mp.mpf(2) / (2 * n + 1))
    lines.append('    {%s, %s},' % (h.hex(), l.hex()))
ln2 = dd(mp.log(2))
hdr = open('logcr_template.hpp (next to this script)').read()
hdr = hdr.replace('@NT@', str(NT)).replace('@TABLE@', '\n'.join(lines)).replace('@LN2HI@', ln2[0].hex()).replace('@LN2LO@', ln2[1].hex())
open('log_cr.hpp', 'w').write(hdr)
