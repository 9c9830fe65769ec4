"""
oracle/ -- TEST INFRASTRUCTURE ONLY.  Not part of the product.

CPU restatement (numpy + plain C) of the one binf hot path this repository
accelerates: HMCSampler.sample() -> _leapfrog() -> pdf.gradient()/log_prob()
and the Gibbs conditional loop around it.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything from here, and only as the checker; ``binf_amd`` never does.

PARITY PINNING STATUS (read before trusting a comparison against this):

* ``import binf`` is impossible in the build container: it needs the third-party
  ``csb`` toolbox (reference ``setup.py:25``), which is not installed, not
  vendored and has no pinned version, and most of the package is Python-2
  source.  No substitute ``csb`` module is written or registered anywhere.
* Pinned by the reference's own tests (the plumbing): ``-13.0`` / ``-29.0``
  (``binf/tests/pdf/__init__.py:67,73,88``), ``252.0`` and ``[252., 396.]``
  (``binf/tests/pdf/likelihoods.py:110-119``) and the Gibbs sweep
  ``x == 3.0, y == 18.0`` (``binf/tests/samplers/gibbs.py:104-112``).
* Pinned by OUTPUTS OF THE REFERENCE'S OWN CODE (round 4): ``HMCSampler._leapfrog``
  (``binf/samplers/hmc.py:92-125``).  ``oracle/gen_ref_leapfrog.py`` compiles the
  reference's class from ``hmc.py`` itself with its single csb import statement
  (``hmc.py:10``) dropped from the syntax tree -- nothing supplied in its place;
  ``__init__`` and ``_leapfrog`` use nothing from csb -- and writes
  ``tests/golden/ref_leapfrog_*.npz``.  ``RefHMCSampler._leapfrog`` and
  ``oracle_c.c`` reproduce those files bit for bit (``tests/test_ref_leapfrog.py``).
* Also pinned by reference-run outputs (round 4, ``oracle/gen_ref_leapfrog.py`` /
  ``oracle/gen_ref_example.py``): ``_adapt_timestep`` (``hmc.py:183-191``), the
  sampler's attribute plumbing, and the EXAMPLE's arithmetic -- the method bodies of
  ``binf/example/likelihood.py:24-30,54-61`` and ``priors.py:23-25,49-54`` compiled
  unchanged and called with a data-only ``self``, class ``RWMCSampler``
  (``samplers.py:54-92``) as it stands, ``GammaSampler``'s shape / rate / sample
  (``samplers.py:27-51``; only the Python-2-only ``_get_prior`` replaced) -- incl. two
  whole chains of ``example_script.py`` that ``ref_example.example_script_chain``
  reproduces bit for bit (``tests/test_ref_example.py``).
* ``HMCSampler.sample`` (``hmc.py:136-164``) is pinned the same way for every statement
  but one: ``oracle/gen_ref_leapfrog.py:_split_sample`` compiles the statements before
  (``:143-150``) and after (``:153-164``) the csb line ``acc = np.random.uniform() <
  exp(-(E_after - E_before))`` (``:151``) unchanged as two methods of the reference class;
  the accept flag passed in between is ``u < numpy.exp(...)`` of the reference's own
  energies.  ``tests/golden/ref_sample_*.npz``; ``RefHMCSampler.sample`` and ``oracle_c.c``
  reproduce energies, flags, states, step sizes and counters bit for bit.
* Still **parity unpinned**: the DEFINITION of ``csb.numeric.exp`` (restated as
  ``numpy.exp(numpy.clip(x, -308, 709))`` from CSB's published source from memory; the
  clip matters only for ``|dE|`` beyond the bounds).  The other golden ``.npz`` files
  under ``tests/golden/`` (``gauss_*``, ``poly_*``, ``dist_*``) are outputs of THIS
  restatement (``oracle/gen_golden.py``), not of the reference.
"""
