"""Numpy model: ARK4(3)6L[2]SA (Kennedy & Carpenter 2003; 6 stages, order 4, embedded order 3, L-stable ESDIRK implicit part) used as a
LINEARLY IMPLICIT additive method on the network ODE:   y' = [f(y) - A y] (explicit tableau) + A y (implicit tableau),
A = the per-protein BLOCK-DIAGONAL part of the Jacobian at the step start (any fixed matrix keeps the order: the additive order
conditions hold for every splitting).  Each implicit stage is ONE block solve with (I - h gamma A) -- the cost of a Rosenbrock-W stage.
Compared against the production ROS34PW2-W (tools/proto_rosw_network.py) on the golden networks: steps and band error (dev tool).
The table is verified against all additive order conditions up to order 4 in exact rational arithmetic (tools/check_ark436.py)."""
import numpy as np, sys, glob
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from fractions import Fraction as Fr
from oracle import network_models as nm
import proto_rosw_network as rw
from check_ark436 import AI, AE, b, bh, c

AIf = np.array([[float(x) for x in r] for r in AI]); AEf = np.array([[float(x) for x in r] for r in AE])
bf = np.array([float(x) for x in b]); bhf = np.array([float(x) for x in bh]); GAM = 0.25

def solve(net, p, y0, t_eval, rtol, atol, norm="max"):
    mask = rw.block_mask(net)
    stops = np.unique(np.concatenate([t_eval[1:], net.kin_grid[(net.kin_grid > t_eval[0]) & (net.kin_grid < t_eval[-1])]]))
    y = y0.copy(); out = np.empty((len(t_eval), net.S)); out[0] = y
    tc = t_eval[0]; I = np.eye(net.S); nst = nrej = 0
    f = lambda yy, tt: nm.rhs(net, p, yy, tt)
    h = 1e-3
    for te in stops:
        while True:
            last = tc + 1.0001*h >= te
            hs = te - tc if last else (0.5*(te-tc) if tc + 2*h > te else h)
            tb = tc
            A = rw.jac_cd(net, p, y, tb) * mask
            Minv = np.linalg.inv(I - hs*GAM*A)
            F = []; G = []                       # F_j = f(Y_j), G_j = A Y_j ;  explicit part N_j = F_j - G_j
            Y = y.copy(); F.append(f(Y, tb)); G.append(A @ Y)
            for i in range(1, 6):
                r = y + hs*sum(AEf[i, j]*(F[j] - G[j]) + AIf[i, j]*G[j] for j in range(i))
                Y = Minv @ r
                F.append(f(Y, tb)); G.append(A @ Y)
            yn = y + hs*sum(bf[j]*F[j] for j in range(6))
            e = hs*sum((bf[j] - bhf[j])*F[j] for j in range(6))
            q = np.abs(e)/(atol + rtol*np.maximum(np.abs(y), np.abs(yn)))
            err = np.max(q) if norm == "max" else np.sqrt(np.mean(q*q))
            fac = max(1/6, min(5, err**(1/4)/0.9)); hnew = hs/fac; nst += 1
            if err <= 1:
                y = yn; tc += hs
                if last:
                    tc = te; h = max(hnew, h) if hs < h else hnew; break
                h = hnew
            else:
                nrej += 1; h = hnew
        idx = np.where(t_eval == te)[0]
        if idx.size: out[idx[0]] = y
    return out, nst, nrej

if __name__ == '__main__':
    files = sys.argv[1:] or sorted(glob.glob('tests/golden/network_m*_small.npz'))
    for fn in files:
        g = np.load(fn); net = nm.Network.from_npz(g)
        for k in (0, 1):
            p = nm.Params.from_npz(g, k)
            for rtol, atol in ((1e-6, 1e-8), (1e-7, 1e-9), (1e-8, 1e-8)):
                Ya, na, ra = solve(net, p, g['y0'], g['t_eval'], rtol, atol)
                Yr, nr, rr = rw.solve(net, p, g['y0'], g['t_eval'], rtol, atol, False)
                print('%-22s set %d %.0e/%.0e: ARK436 %5d steps (%d rej, %d stage solves) band %.4f | ROS34PW2 %5d steps (%d stage solves) band %.4f | ref lsoda8 %.3f' % (
                    fn.split('/')[-1], k, rtol, atol, na, ra, 5*na, rw.band(Ya, g['Y_tight'][k]), nr, 4*nr, rw.band(Yr, g['Y_tight'][k]), rw.band(g['Y_lsoda8'][k], g['Y_tight'][k])), flush=True)


def solve_sgs(net, p, y0, t_eval, rtol, atol):
    """The same method with the combinatorial kernels' APPROXIMATE block factorisation P = (D_g - F) D_g^-1 (D_g - K) in place of g I - A:
    the implicit operator becomes A~ = g I - P = A - F D_g^-1 K (it changes with the step size, which an additive method does not mind);
    the defect F D_g^-1 K is then integrated by the EXPLICIT tableau -- this experiment asks whether its stability limit bites."""
    mask = rw.block_mask(net)
    stops = np.unique(np.concatenate([t_eval[1:], net.kin_grid[(net.kin_grid > t_eval[0]) & (net.kin_grid < t_eval[-1])]]))
    y = y0.copy(); out = np.empty((len(t_eval), net.S)); out[0] = y
    tc = t_eval[0]; I = np.eye(net.S); nst = nrej = 0
    f = lambda yy, tt: nm.rhs(net, p, yy, tt)
    h = 1e-3
    for te in stops:
        while True:
            last = tc + 1.0001*h >= te
            hs = te - tc if last else (0.5*(te-tc) if tc + 2*h > te else h)
            tb = tc
            Ab = rw.jac_cd(net, p, y, tb) * mask
            g = 1.0/(hs*GAM)
            D = np.diag(np.diag(Ab)); F = np.tril(Ab, -1); K = np.triu(Ab, 1)
            Dg = g*I - D
            P = (Dg - F) @ np.linalg.inv(Dg) @ (Dg - K)
            At = g*I - P
            Pinv = np.linalg.inv(P)
            Fs = []; Gs = []
            Y = y.copy(); Fs.append(f(Y, tb)); Gs.append(At @ Y)
            for i in range(1, 6):
                r = y + hs*sum(AEf[i, j]*(Fs[j] - Gs[j]) + AIf[i, j]*Gs[j] for j in range(i))
                Y = Pinv @ (g*r)
                Fs.append(f(Y, tb)); Gs.append(At @ Y)
            yn = y + hs*sum(bf[j]*Fs[j] for j in range(6))
            e = hs*sum((bf[j] - bhf[j])*Fs[j] for j in range(6))
            err = np.max(np.abs(e)/(atol + rtol*np.maximum(np.abs(y), np.abs(yn))))
            fac = max(1/6, min(5, err**(1/4)/0.9)); hnew = hs/fac; nst += 1
            if err <= 1:
                y = yn; tc += hs
                if last:
                    tc = te; h = max(hnew, h) if hs < h else hnew; break
                h = hnew
            else:
                nrej += 1; h = hnew
        idx = np.where(t_eval == te)[0]
        if idx.size: out[idx[0]] = y
    return out, nst, nrej


if __name__ == '__main__' and False:
    pass
