"""Operator layer: thin torch wrappers (autograd Functions) over the C-ABI HIP library."""


def invalidate_caches(module):
    """Drop every cached re-layout of a weight / folded BatchNorm vector under ``module`` (Winograd filters, tap-major and
    transposed bf16 copies, packed stem filters, eval-mode BatchNorm folds).  The caches serve FROZEN tensors only and are
    keyed on (version counter, storage address); call this after anything that rewrites frozen weights without moving
    those -- ``.data`` assignments, an EMA teacher update, a fused optimizer stepping weights that are frozen again
    afterwards.  ``load_state_dict`` / ``copy_`` bump the version counters and need no call.  -> number of entries dropped."""
    n = 0
    seen = set()
    mods = list(module.modules())
    objs = mods + [t for m in mods for t in list(m._parameters.values()) + list(m._buffers.values()) if t is not None]
    for o in objs:
        if id(o) in seen:
            continue
        seen.add(id(o))
        d = getattr(o, "__dict__", None)
        if not d:
            continue
        for k in [k for k in d if isinstance(k, str) and k.startswith("_ud_")]:
            try:
                delattr(o, k)
                n += 1
            except AttributeError:
                pass
    return n
