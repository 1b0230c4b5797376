"""Dev tool: time solve kernels for several (model, n_sites, B, method, linsolve) combos via pk_time_solve_protein_batch."""
import sys, ctypes as C, time
import numpy as np, torch
import pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from phoskintime_amd import batch, _capi

def run(model, n, B, method, lin, lo=0.0, hi=20.0, rtol=1e-7, atol=1e-9, iters=3, rk4_h=None, seed=20260517):
    ctx = batch.get_context()
    dev = torch.device('cuda', ctx.device)
    P, S = batch.n_params(model, n), batch.n_states(model, n)
    rng = np.random.default_rng(seed)
    th = torch.as_tensor(rng.uniform(lo, hi, (B, P)), device=dev)
    y0 = torch.ones(S, dtype=torch.float64, device=dev)
    t = torch.as_tensor(np.array([0.0, 0.5, 0.75, 1.0, 2.0, 4.0, 8.0, 16.0, 30.0, 60.0, 120.0, 240.0, 480.0, 960.0]), device=dev)
    T = t.numel()
    sol = torch.empty((B, T, S), dtype=torch.float64, device=dev)
    st = torch.zeros(B, dtype=torch.int32, device=dev); ns = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    opts = _capi.default_opts(method=method, linsolve=lin, rtol=rtol, atol=atol, rk4_h=rk4_h)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    p = lambda x: C.c_void_p(x.data_ptr())
    args = (model, n, B, p(th), p(y0), 0, p(t), T, C.byref(opts), p(sol), None, None, 0, p(st), p(ns))
    ms = ctx.lib.pk_time_solve_protein_batch(ctx.handle, 1, *args)   # warm-up
    ms = ctx.lib.pk_time_solve_protein_batch(ctx.handle, iters, *args)
    nsc = ns.cpu().numpy(); stc = st.cpu().numpy()
    print('model %d n %2d S %2d B %6d %-6s %-10s U(%g,%g) rtol %.0e: %9.3f ms/launch  %12.0f replicas/s  steps mean %.0f max %d  rej mean %.1f  bad %d' % (
        model, n, S, B, method, lin, lo, hi, rtol, ms, B / (ms * 1e-3), nsc[:, 0].mean(), nsc[:, 0].max(), nsc[:, 1].mean(), (stc != 0).sum()), flush=True)
    return ms

if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if which in ('all', 'c3', 'lrp'):
        for meth in ('lrp8', 'rodas4'):
            for lin in ('auto', 'structured', 'dense'):
                run(0, 30, 65536 if lin != 'dense' else 16384, meth, lin)
            run(1, 14, 65536, meth, 'structured')
            run(1, 14, 16384, meth, 'dense')
            run(2, 4, 16384, meth, 'dense')
            run(2, 5, 4096, meth, 'dense')
            for nb in (1, 2, 3, 4, 5):
                run(2, nb, 65536, meth, 'auto')
            run(0, 4, 65536, meth, 'auto')
    if which in ('all', 'lrp12'):
        for meth, rt, at in (('lrp8', 1e-7, 1e-9), ('lrp12', 1e-6, 1e-8), ('lrp12', 1e-7, 1e-9), ('lrp8', 1e-6, 1e-8)):
            run(0, 30, 65536, meth, 'auto', rtol=rt, atol=at)
            run(0, 4, 65536, meth, 'auto', rtol=rt, atol=at)
            run(1, 14, 65536, meth, 'auto', rtol=rt, atol=at)
            run(2, 4, 65536, meth, 'auto', rtol=rt, atol=at)
            run(2, 5, 65536, meth, 'auto', rtol=rt, atol=at)
    if which == 'layouts':
        for n in (4, 8, 10, 12, 14, 16, 18, 20, 24, 28, 30, 32, 40, 48, 56, 62):
            run(0, n, 65536, 'lrp12', 'auto', rtol=1e-6, atol=1e-8, iters=20)
    if which == 'tprB':
        import os
        for model, n in ((0, 4), (0, 12), (1, 4), (1, 14)):
            for Bx in (1024, 4096, 8192, 16384, 32768):
                for v in ('0', '1'):
                    os.environ['PK_TPR'] = v
                    print('PK_TPR=' + v, end='  ')
                    run(model, n, Bx, 'lrp12', 'auto', rtol=1e-6, atol=1e-8, iters=20)
    if which == 'tpr':
        import os
        for model, ns in ((2, (1, 2, 3)),) if len(sys.argv) > 2 and sys.argv[2] == 'rand' else ((0, (1, 4, 8, 12)), (1, (1, 4, 8, 14))):
            for n in ns:
                for Bx in (65536, 524288):
                    for v in ('0', '1'):
                        os.environ['PK_TPR'] = v
                        print('PK_TPR=' + v, end='  ')
                        run(model, n, Bx, 'lrp12', 'auto', rtol=1e-6, atol=1e-8, iters=10)
    if which in ('all', 'rand'):
        for nb in (1, 2, 3, 4, 5, 6):
            run(2, nb, 65536 if nb < 6 else 16384, 'lrp12', 'auto', rtol=1e-6, atol=1e-8, iters=10)
    if which in ('all', 'c3'):
        run(0, 30, 65536, 'rodas4', 'structured', 0.05, 2.0)
        run(0, 30, 65536, 'bdf2', 'structured')
        run(0, 30, 8192, 'bdf2', 'dense')
    if which in ('all', 'c2'):
        run(1, 14, 4096, 'lrp8', 'auto')
        run(1, 14, 4096, 'lrp8', 'auto', 0.05, 2.0)
        run(1, 14, 4096, 'rodas4', 'structured')
        run(1, 14, 4096, 'bdf2', 'structured')
        run(1, 14, 65536, 'lrp8', 'auto')
        run(1, 14, 4096, 'rk4', 'auto', 0.05, 2.0, rk4_h=0.02)
        run(1, 14, 4096, 'rk4', 'auto', 0.0, 20.0, rk4_h=0.01)
    if which in ('all', 'misc'):
        run(0, 4, 65536, 'rodas4', 'structured')
        run(0, 4, 65536, 'rodas4', 'dense')
        run(2, 4, 16384, 'rodas4', 'dense')
        run(2, 5, 4096, 'rodas4', 'dense')
        run(2, 3, 65536, 'rodas4', 'dense')
