"""
Checkpoint / resume of a sampling run (not in the reference, whose samples are a Python list
in memory, ``example_script.py:32-34``): the state every sampler, generator and sample store
needs to go on EXACTLY where it stopped -- a resumed run draws bit for bit what the
uninterrupted run draws (``tests/test_gpu_checkpoint.py``).  Possible because every device
draw is a function of ``(seed, stream position, global chain index)`` and the samplers' whole
memory is a few tensors (state, per-chain step sizes, counters).

    ckpt = checkpoint.state_dict(gibbs=gips, store=store)      # tensors moved to the host
    checkpoint.save('run.pt', gibbs=gips, store=store)         # torch.save of the same
    ...
    checkpoint.load('run.pt', gibbs=gips2, store=store2)       # into freshly built objects

Only tensors, numbers, strings, lists and dicts are written: ``load`` reads the file with
``torch.load(..., weights_only=True)``.  Objects take part by having ``state_dict()`` /
``load_state_dict(d)``: ``HMCSampler``, ``GibbsSampler``, ``RWMCSampler``, ``GammaSampler``,
``DeviceRNG``, ``HostLegacyRNG`` (the global legacy numpy stream), ``SampleStore``.
"""
import torch


def _to_host(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu()
    if isinstance(x, dict):
        return {k: _to_host(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_to_host(v) for v in x]
    return x


def state_dict(**objects):
    """``{name: object.state_dict()}`` with every tensor copied to the host."""
    return {name: _to_host(obj.state_dict()) for name, obj in objects.items()}


def load_state_dict(ckpt, **objects):
    for name, obj in objects.items():
        if name not in ckpt:
            raise KeyError('checkpoint has no entry %r (it holds %s)' % (name, sorted(ckpt)))
        obj.load_state_dict(ckpt[name])


def save(path, **objects):
    torch.save(state_dict(**objects), path)


def load(path, **objects):
    """Read ``path`` (tensors and plain containers only) and hand every named object its entry;
    returns the whole checkpoint dict."""
    ckpt = torch.load(path, map_location='cpu', weights_only=True)
    load_state_dict(ckpt, **objects)
    return ckpt


def like(value, reference):
    """``value`` from a checkpoint on the device (and with the dtype) of ``reference``."""
    if isinstance(value, torch.Tensor) and isinstance(reference, torch.Tensor):
        return value.to(device=reference.device, dtype=reference.dtype).contiguous()
    return value
