"""
TEST INFRASTRUCTURE ONLY.  Writes tests/golden/ref_api_surface.json: the NAMES of the classes,
functions and methods every hot-path module of the reference defines (SURVEY section 2's starred
rows plus the example application) -- names only, read off the source text with two regular
expressions (several reference files are Python-2 source and do not parse).  The mirror package
must offer every one of them (tests/test_host_mirror.py::test_mirror_offers_every_name...), so a
binf program finds what it imports after ``s/binf/binf_amd/``.

Run (build container only):   python -m oracle.gen_ref_surface
"""
import json
import os
import re

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden',
                   'ref_api_surface.json')
MODULES = {
    'binf': 'binf/__init__.py', 'binf.samplers': 'binf/samplers/__init__.py',
    'binf.samplers.hmc': 'binf/samplers/hmc.py', 'binf.samplers.gibbs': 'binf/samplers/gibbs.py',
    'binf.pdf': 'binf/pdf/__init__.py', 'binf.pdf.posteriors': 'binf/pdf/posteriors.py',
    'binf.pdf.likelihoods': 'binf/pdf/likelihoods.py', 'binf.pdf.priors': 'binf/pdf/priors.py',
    'binf.model': 'binf/model/__init__.py', 'binf.model.forwardmodels': 'binf/model/forwardmodels.py',
    'binf.model.errormodels': 'binf/model/errormodels.py',
    'binf.example.likelihood': 'binf/example/likelihood.py', 'binf.example.priors': 'binf/example/priors.py',
    'binf.example.misc': 'binf/example/misc.py', 'binf.example.samplers': 'binf/example/samplers.py'}


def main():
    surface = {}
    for mod, path in MODULES.items():
        names, cls = {}, None
        for line in open(os.path.join(REF, path)).read().split('\n'):
            m = re.match(r'class (\w+)', line)
            if m:
                cls = m.group(1)
                names[cls] = []
                continue
            m = re.match(r'def (\w+)', line)
            if m:
                cls = None
                names[m.group(1)] = None
                continue
            m = re.match(r'\s+def (\w+)', line)
            if m and cls:
                names[cls].append(m.group(1))
        surface[mod] = names
    json.dump({'provenance': 'names defined by the reference modules listed (regular expressions over '
                             'the source text, oracle/gen_ref_surface.py); names only',
               'modules': surface}, open(OUT, 'w'), indent=1, sort_keys=True)
    print('wrote', OUT, sum(1 + len(v or []) for d in surface.values() for v in d.values()), 'names')


if __name__ == '__main__':
    main()
