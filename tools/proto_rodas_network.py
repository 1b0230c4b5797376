"""Numpy experiment (dev tool): does a classical order-4 Rosenbrock method (RODAS4, 6 stages) keep its advantage over the order-3
W-method ROS34PW2 when the network integrator hands it only the per-protein BLOCK-DIAGONAL Jacobian (an inexact Jacobian costs a
classical Rosenbrock method formal order, a W-method none)?  Prints steps and band error vs the reference's LSODA@1e-12."""
import glob, sys
import numpy as np
sys.path.insert(0, '.')
from oracle import network_models as nm
from tools.proto_rosw_network import block_mask, jac_cd, band, solve as solve_rosw

G4 = 0.25
A4 = {2: [0.1544e+01], 3: [0.9466785280815826, 0.2557011698983284], 4: [0.3314825187068521e+01, 0.2896124015972201e+01, 0.9986419139977817],
      5: [0.1221224509226641e+01, 0.6019134481288629e+01, 0.1253708332932087e+02, -0.6878860361058950]}
C4 = {2: [-0.56688e+01], 3: [-0.2430093356833875e+01, -0.2063599157091915], 4: [-0.1073529058151375, -0.9594562251023355e+01, -0.2047028614809616e+02],
      5: [0.7496443313967647e+01, -0.1024680431464352e+02, -0.3399990352819905e+02, 0.1170890893206160e+02],
      6: [0.8083246795921522e+01, -0.7981132988064893e+01, -0.3152159432874371e+02, 0.1631930543123136e+02, -0.6058818238834054e+01]}


def solve_rodas4(net, p, y0, t_eval, rtol, atol, full_jac=False):
    mask = block_mask(net)
    stops = np.unique(np.concatenate([t_eval[1:], net.kin_grid[(net.kin_grid > t_eval[0]) & (net.kin_grid < t_eval[-1])]]))
    y = y0.copy(); out = np.empty((len(t_eval), net.S)); out[0] = y
    tc = t_eval[0]; I = np.eye(net.S); nst = nrej = 0
    f = lambda yy, tt: nm.rhs(net, p, yy, tt)
    h = 1e-3
    for te in stops:
        while True:
            last = tc + 1.0001 * h >= te
            hs = te - tc if last else (0.5 * (te - tc) if tc + 2 * h > te else h)
            tb = tc
            J = jac_cd(net, p, y, tb)
            if not full_jac: J = J * mask
            Winv = np.linalg.inv(I / (G4 * hs) - J)
            U = [Winv @ f(y, tb)]
            for i in (2, 3, 4):
                Y = y + sum(a * u for a, u in zip(A4[i], U))
                U.append(Winv @ (f(Y, tb) + sum(c * u for c, u in zip(C4[i], U)) / hs))
            yn = y + sum(a * u for a, u in zip(A4[5], U))
            U.append(Winv @ (f(yn, tb) + sum(c * u for c, u in zip(C4[5], U)) / hs))
            yn = yn + U[4]
            U.append(Winv @ (f(yn, tb) + sum(c * u for c, u in zip(C4[6], U)) / hs))
            yn = yn + U[5]
            err = np.max(np.abs(U[5]) / (atol + rtol * np.maximum(np.abs(y), np.abs(yn))))
            fac = max(1 / 6, min(5, err ** 0.25 / 0.9)); hnew = hs / fac; nst += 1
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
    for fn in ['tests/golden/network_m0_small.npz', 'tests/golden/network_m4_small.npz', 'tests/golden/network_m0_medium.npz']:
        g = np.load(fn); net = nm.Network.from_npz(g)
        for k in (0, 1):
            p = nm.Params.from_npz(g, k)
            for rtol, atol in ((1e-5, 1e-7), (1e-7, 1e-9)):
                Yw, nw, rw = solve_rosw(net, p, g['y0'], g['t_eval'], rtol, atol, False)
                print('%-24s set %d rtol %.0e ROS34PW2 blockJ: band %.4f steps %d rej %d  (stage evals %d)' % (fn.split('/')[-1], k, rtol, band(Yw, g['Y_tight'][k]), nw, rw, 4 * nw), flush=True)
                for full in (False, True):
                    Y, nst, nrej = solve_rodas4(net, p, g['y0'], g['t_eval'], rtol, atol, full)
                    print('%-24s set %d rtol %.0e RODAS4 %s: band %.4f steps %d rej %d  (stage evals %d)' % (
                        fn.split('/')[-1], k, rtol, 'fullJ ' if full else 'blockJ', band(Y, g['Y_tight'][k]), nst, nrej, 6 * nst), flush=True)
