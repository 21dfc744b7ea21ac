"""Branch pinning for parity tests of piecewise-linear nets (see tests/test_step_b64_gpu.py for the why).

record_branches(out)    taps every fused lrelu / relu of the HIP path at the t2i_amd.kernels wrappers and appends
                        (activation output > 0) to `out`, in launch order;
split_sections(...)     cuts that flat list into the oracle's named sections (one per network pass), using the section
                        sizes and shapes of an oracle run; a HIP pass that batches several oracle passes (the critics
                        without batch norm run D(G), D(x), D(x_mismatch) as one pass over 3B samples) is split along the batch."""
import contextlib


@contextlib.contextmanager
def record_branches(out):
    from t2i_amd import kernels as K
    saved = {}

    def wrap(name, act_pos, first=False):
        fn = getattr(K, name)
        saved[name] = fn

        def tapped(*a, **kw):
            y = fn(*a, **kw)
            if first:                      # (the activation output is the first element of a tuple)
                res, y = y, y[0]
            act = a[act_pos] if len(a) > act_pos else kw.get('act', K.ACT_NONE)
            if act in (K.ACT_LRELU, K.ACT_RELU):
                out.append((y > 0).cpu())
            return res if first else y
        setattr(K, name, tapped)

    try:
        wrap('conv_fwd', 5); wrap('conv_fwd_stats', 5); wrap('conv_bwd_data', 5)
        wrap('bn_train_fwd_grouped', 6, first=True); wrap('bn_apply_groups', 3); wrap('bn_apply', 3); wrap('add_act', 2); wrap('act_fwd', 1)
        yield out
    finally:
        for n, fn in saved.items():
            setattr(K, n, fn)


def to_oracle_layout(m, like):
    """HIP activations are NHWC (dense layers run as 1x1 convs on [B,1,1,C]); the oracle's are NCHW / [B,C]."""
    if m.dim() == 4 and like.dim() == 2:
        return m.reshape(m.shape[0], -1)
    if m.dim() == 2 and like.dim() == 2:
        return m
    return m.permute(0, 3, 1, 2).contiguous()


def split_sections(rec, oracle_record, plan):
    """rec: flat HIP list; oracle_record: {section: [masks]} of an oracle run of the same step; plan: the HIP passes in
    launch order, each a tuple of the oracle sections it covers (more than one = batched along axis 0).
    -> {section: [masks in oracle layout]}"""
    out, pos = {}, 0
    for group in plan:
        n = len(oracle_record[group[0]])
        assert all(len(oracle_record[s]) == n for s in group), group
        for i in range(n):
            like = oracle_record[group[0]][i]
            m = to_oracle_layout(rec[pos], like)
            pos += 1
            B = like.shape[0]
            assert m.shape[0] == B * len(group) and m.shape[1:] == like.shape[1:], (group, i, tuple(m.shape), tuple(like.shape))
            for j, s in enumerate(group):
                out.setdefault(s, []).append(m[j * B:(j + 1) * B])
    assert pos == len(rec), (pos, len(rec))
    return out


def flips(oracle_record, masks):
    n = f = 0
    for k, rec in oracle_record.items():
        for a, b in zip(rec, masks[k]):
            n += a.numel()
            f += int((a != b).sum())
    return f, n
