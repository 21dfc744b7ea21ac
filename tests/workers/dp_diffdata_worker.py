"""Worker of tests/test_dp_diffdata_gpu.py: the data-parallel CONTRACT on different data, two gloo ranks sharing GPU 0.

north_star / SURVEY 8(e): "N replicas at local batch b behave as the reference at BATCH_SIZE N*b" holds for the critic because it
has no batch coupling — per-sample gradient penalty (reference models/wgancls/model.py:63-65), no batch norm in d_net, every
loss a batch mean (:85, :100) — and for the generator only per replica (its batch norm uses the replica's own statistics).
What is checked, with DISTINCT feeds on the two ranks (local batch b each):

 critic step (fake images handed in, so that the generator's per-replica batch norm is not part of the comparison):
   (1) the exchanged arena holds EXACTLY local_0 + local_1, where local_r is what a single process computes on rank r's feed;
   (2) its mean equals the gradient of a SINGLE PROCESS AT BATCH 2b on the concatenated feed to fp32 summation-order accuracy
       (relative L2 <= 2e-5 per tensor, measured ~1e-6), and so does the global-batch kt step (|dkt| <= 1e-6) and every logged
       batch-mean scalar once averaged over the ranks;
 generator step:
   (3) the exchanged arena holds EXACTLY local_0 + local_1 — each rank is the reference at batch b (per-replica batch norm), the
       update uses the mean of the two gradients; it is NOT the batch-2b gradient, and the test states the difference it measures.
Rank 0 prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def make_cfg(batch):
    import bench
    cfg = bench.make_cfg(batch)
    cfg.MODEL.GF_DIM = 32          # quarter width: seconds instead of a minute, same graph
    cfg.MODEL.DF_DIM = 32
    return cfg


def full_feed(cfg2, device):
    import bench
    f = bench.synthetic_feed(cfg2, device, seed=4242)
    g = torch.Generator(device=device).manual_seed(99)
    f['fake'] = torch.rand(f['x'].shape, generator=g, device=device) * 2 - 1       # stands in for G(z) in the critic step
    return f


def part(feed, lo, hi):
    return {k: (v[lo:hi].contiguous() if torch.is_tensor(v) and v.dim() > 0 else v) for k, v in feed.items()}


def critic_only(model, feed):
    """d_losses with the fake images taken from the feed (the generator's batch norm stays out of the picture)."""
    fake = feed['fake']
    orig = model.generator
    model.generator = lambda z, cond, reuse=False, **kw: (fake, None, None)
    try:
        return model.d_losses(feed)
    finally:
        model.generator = orig


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    assert world == 2
    dist.init_process_group('gloo')
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    import t2i_amd  # noqa: F401
    from t2i_amd.dp import DataParallel
    from t2i_amd.models.wgancls.model import WGanCls
    b = 4
    cfg, cfg2 = make_cfg(b), make_cfg(2 * b)
    whole = full_feed(cfg2, dev)
    mine = part(whole, rank * b, (rank + 1) * b)
    report = {}

    # ---- single-process references on this rank's own feed (local_r) --------------------------------------------------------
    m = WGanCls(cfg, device=dev, seed=0)
    out_l = critic_only(m, mine)
    d_local = m.d_arena.grad.clone()
    wd_local = out_l['wd_sums'].clone()
    m.g_losses(mine)
    g_local = m.g_arena.grad.clone()
    del m

    # ---- the two-rank run ----------------------------------------------------------------------------------------------
    dp = DataParallel(bucket_bytes=1 << 20)
    m = WGanCls(cfg, device=dev, seed=0, dp=dp)
    dp.broadcast_variables(m.store)
    kt0 = float(m.kt)
    out = critic_only(m, mine)
    scale = dp.allreduce_arena(m.d_arena, extra=out['wd_sums'])
    torch.cuda.synchronize()
    d_sum = m.d_arena.grad.clone()
    wd_sum = out['wd_sums'].clone()
    def gathered(t):                      # gloo gathers host tensors
        parts = [torch.empty_like(t, device='cpu') for _ in range(2)]
        dist.all_gather(parts, t.cpu())
        return parts
    others = gathered(d_local)
    report['critic_sum_is_exactly_local0_plus_local1'] = bool(torch.equal(d_sum.cpu(), others[0] + others[1]))
    wds = gathered(wd_local)
    report['kt_means_sum_exact'] = bool(torch.equal(wd_sum.cpu(), wds[0] + wds[1]))
    scal_keys = ('D_loss', 'wdist', 'wdist2', 'real_gp', 'real_gp2')
    scal = torch.stack([out[k].detach().double().reshape(()) for k in scal_keys]).cpu()
    dist.all_reduce(scal)
    scal = (scal / 2).tolist()
    # the generator step against the SAME (not yet updated) critic the single-process references above used
    m.g_losses(mine)
    dp.allreduce_arena(m.g_arena)
    torch.cuda.synchronize()
    g_sum = m.g_arena.grad.clone()
    gothers = gathered(g_local)
    report['generator_sum_is_exactly_local0_plus_local1'] = bool(torch.equal(g_sum.cpu(), gothers[0] + gothers[1]))
    m.D_optim.prepare(1e-4)
    m._d_update(out, scale)
    torch.cuda.synchronize()
    kt_dp = float(m.kt)
    offsets = dict(m.d_arena.offsets)
    g_offsets = dict(m.g_arena.offsets)
    del m

    # ---- rank 0: ONE process at batch 2b on the concatenated feed -----------------------------------------------------------
    if rank == 0:
        m2 = WGanCls(cfg2, device=dev, seed=0)
        out2 = critic_only(m2, whole)
        torch.cuda.synchronize()
        d2 = m2.d_arena.grad.clone()
        worst, worst_name = 0.0, None
        for n, (o, k) in offsets.items():
            ref = d2[o:o + k]
            if float(ref.abs().max()) == 0.0:
                continue
            r = rel(d_sum[o:o + k] / 2, ref)
            if r > worst:
                worst, worst_name = r, n
        report['critic_mean_vs_batch_2b_worst_rel_l2'] = worst
        report['critic_worst_tensor'] = worst_name
        m2.g_losses(whole)                       # (before the critic update, like the two-rank run)
        torch.cuda.synchronize()
        g2 = m2.g_arena.grad.clone()
        m2.D_optim.prepare(1e-4)
        m2._d_update(out2, 1.0)
        torch.cuda.synchronize()
        report['kt_start'] = kt0
        report['kt_dp'] = kt_dp
        report['kt_batch_2b'] = float(m2.kt)
        report['scalars_mean_over_ranks'] = dict(zip(scal_keys, scal))
        report['scalars_batch_2b'] = {k: float(out2[k]) for k in scal_keys}
        # the generator: per-replica batch norm makes the two-rank mean differ from the batch-2b gradient (stated, not demanded)
        worst_g = 0.0
        for n, (o, k) in g_offsets.items():
            ref = g2[o:o + k]
            if float(ref.abs().max()) == 0.0:
                continue
            worst_g = max(worst_g, rel(g_sum[o:o + k] / 2, ref))
        report['generator_mean_vs_batch_2b_worst_rel_l2__not_a_contract'] = worst_g
        print('DPDIFF ' + json.dumps(report), flush=True)
    else:
        flags = {k: v for k, v in report.items() if isinstance(v, bool)}
        assert all(flags.values()), flags
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
