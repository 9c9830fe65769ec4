"""
Device buffers of the per-chain chi^2 memos (``binf_poly_gauss_logp_memo_f64``,
``binf_pairdist_gauss_logp_memo_f64``, ``binf_pairdist_hmc_energy_f64``).

A memo belongs to a DATA SET (the device copy of the error model's ``ys``), a batch
shape and a stream.  The cache is keyed WEAKLY by the data tensor: when the last
model holding that data goes away, so do its memos (they used to sit in
module-level dicts for the life of the process).  What is alive is capped by BYTES
(``MAX_BYTES``, least recently used first): one pair-distance memo is up to 512 MiB
of HBM (two copies of the ``[C x 3n]`` coordinates), so an entry count says little.
``clear_chi2_memos()`` drops everything at once.
"""
import collections
import weakref

MAX_BYTES = 2 << 30          # HBM held by memos, all data sets together

# id(data tensor) -> (weak reference to it, OrderedDict(key -> (other, memo, bytes, [last use])));
# the weak reference's callback removes the entry with the tensor (ids are compared, never
# tensors: a tensor's == is elementwise)
_by_data = {}
_clock = [0]


def _entries():
    for ident, (_, od) in list(_by_data.items()):
        for key, ent in list(od.items()):
            yield ident, od, key, ent


def bytes_held():
    return sum(ent[2] for _, _, _, ent in _entries())


def clear_chi2_memos():
    """Release every memo buffer (they are rebuilt, all-miss, when next needed)."""
    for _, od in list(_by_data.values()):
        od.clear()


def chi2_memo(data, other, shape, device, make):
    """The memo ``(args [2 x C x K], chi2 [2 x C], state [2 x C])`` for data tensor
    ``data`` (weak key), the second tensor that defines the data set ``other`` (``xs``
    or the pair index: identity is checked), batch ``shape`` and the current stream of
    ``device``; ``make(C, K, device)`` builds a fresh one."""
    from binf_amd import _native
    C, K = int(shape[0]), int(shape[1])
    key = (id(other), C, K, _native.stream_handle(device))
    slot = _by_data.get(id(data))
    if slot is None or slot[0]() is not data:
        ident = id(data)
        slot = (weakref.ref(data, lambda _r, ident=ident: _by_data.pop(ident, None)),
                collections.OrderedDict())
        _by_data[ident] = slot
    od = slot[1]
    ent = od.get(key)
    if ent is not None and ent[0]() is other:
        od.move_to_end(key)
        _clock[0] += 1
        ent[3][0] = _clock[0]
        return ent[1]
    memo = make(C, K, device)
    nbytes = sum(t.numel() * t.element_size() for t in memo)
    _clock[0] += 1
    od[key] = (weakref.ref(other), memo, nbytes, [_clock[0]])
    # least recently used entries (of any data set) go first; the new one stays
    while bytes_held() > MAX_BYTES:
        oldest = min(((e[3][0], d, o, k) for d, o, k, e in _entries() if e[1] is not memo),
                     key=lambda t: t[0], default=None)
        if oldest is None:
            break
        oldest[2].pop(oldest[3], None)
    return memo
