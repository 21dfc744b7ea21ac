import torch, time
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for mb in (8, 33.5, 100, 400):
    n = int(mb * 1e6 / 4)
    a = torch.empty(n, device='cuda'); b = torch.empty(n, device='cuda')
    print('%6.1f MB  fill %7.1f us (%.2f TB/s)  copy %7.1f us (%.2f TB/s rd+wr)  sum %7.1f us (%.2f TB/s)' % (
        mb, t(lambda: a.fill_(1.0)), mb / t(lambda: a.fill_(1.0)), t(lambda: b.copy_(a)), 2 * mb / t(lambda: b.copy_(a)),
        t(lambda: a.sum()), mb / t(lambda: a.sum())))
