"""L-stable restricted-Pade one-step methods for AFFINE systems y' = J y + b (dev tool; DESIGN.md "LRP").

  M = I - gamma h J,  z_1 = M^{-1} h f(y_n),  z_{k+1} = M^{-1} z_k  (s solves, ONE factorisation, ONE rhs)
  y_{n+1} = y_n + sum_k beta_k z_k   <=>  stability function R(z) = 1 + sum_k beta_k z / (1 - gamma z)^k

beta is fixed by: order p = s - 1 (R(z) = exp(z) + O(z^{p+1})) and L-stability R(inf) = 0 (beta_1 = gamma).
The embedded estimator uses a second weight set of order p - 1 with R_hat(inf) = 0 on the same z_k.
gamma must lie in the A-stability window of Hairer & Wanner II, Table IV.6.4; A-stability is re-checked numerically here."""
import mpmath as mp
mp.mp.dps = 60

def series_coeffs(s, gamma, nterms):
    """c[k][m] = coefficient of z^m in z/(1-gamma z)^k, m = 0..nterms-1."""
    out = []
    for k in range(1, s + 1):
        # (1 - g z)^-k = sum_j C(k+j-1, j) g^j z^j
        row = [mp.mpf(0)] * nterms
        for j in range(nterms - 1):
            row[j + 1] = mp.binomial(k + j - 1, j) * gamma ** j
        out.append(row)
    return out

def solve_weights(s, gamma, order, extra_zero=()):
    """beta with R(z) - exp(z) = O(z^{order+1}), beta_1 = gamma (L-stable), remaining freedom: beta_k = 0 for k in extra_zero."""
    n = order + 1
    c = series_coeffs(s, gamma, n)
    rows, rhs = [], []
    for m in range(1, n):                      # z^m coefficient must equal 1/m!
        rows.append([c[k][m] for k in range(s)]); rhs.append(1 / mp.factorial(m))
    rows.append([mp.mpf(1)] + [mp.mpf(0)] * (s - 1)); rhs.append(gamma)          # beta_1 = gamma
    for k in extra_zero:
        r = [mp.mpf(0)] * s; r[k - 1] = 1; rows.append(r); rhs.append(mp.mpf(0))
    A = mp.matrix(rows); b = mp.matrix(rhs)
    if A.rows == A.cols:
        x = mp.lu_solve(A, b)
    else:
        x = mp.qr_solve(A, b)[0]
    return [x[i] for i in range(s)]

def R(beta, gamma, z):
    return 1 + sum(beta[k] * z / (1 - gamma * z) ** (k + 1) for k in range(len(beta)))

def a_stable(beta, gamma, n=4000):
    worst = 0
    for i in range(1, n):
        y = mp.mpf(10) ** (mp.mpf(i) / n * 8 - 4)          # 1e-4 .. 1e4 on the imaginary axis
        worst = max(worst, abs(R(beta, gamma, mp.mpc(0, y))))
    return worst

if __name__ == '__main__':
    import sys
    s = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    for gs in (sys.argv[2:] or ['0.2', '0.25', '0.3']):
        g = mp.mpf(gs)
        beta = solve_weights(s, g, s - 1)
        # embedded: order s-2, L-stable, beta_s = 0 (uses z_1..z_{s-1})
        bh = solve_weights(s, g, s - 2, extra_zero=(s,))
        # principal error constants
        c = series_coeffs(s, g, s + 2)
        ec = sum(beta[k] * c[k][s] for k in range(s)) - 1 / mp.factorial(s)
        ech = sum(bh[k] * c[k][s - 1] for k in range(s)) - 1 / mp.factorial(s - 1)
        print('s=%d gamma=%s  max|R(iy)|=%s  max|Rhat(iy)|=%s  err const C_%d=%s  Chat_%d=%s' % (s, gs, mp.nstr(a_stable(beta, g), 8), mp.nstr(a_stable(bh, g), 8), s, mp.nstr(ec, 5), s - 1, mp.nstr(ech, 5)))
        print('   beta =', [mp.nstr(x, 22) for x in beta])
        print('   eps  =', [mp.nstr(x - y, 22) for x, y in zip(beta, bh)])
