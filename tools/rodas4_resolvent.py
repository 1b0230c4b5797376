"""RODAS4 for AFFINE right-hand sides f(y) = J y + b in resolvent form (dev tool; derivation in DESIGN.md).

With M = I - gamma h J, z_1 = M^{-1} (h f(y_n)), z_{k+1} = M^{-1} z_k the six Rosenbrock stage vectors are fixed linear
combinations u_i = sum_k Theta[i][k] z_k, because h J = (I - M) / gamma turns every J-product into a shift k -> k + 1:
    u_i = M^{-1}[gamma h f0 + sum_j (a_ij + gamma c_ij) u_j] - sum_j a_ij u_j .
Hence  y_{n+1} = y_n + sum_k beta_k z_k  and  err = u_6 = sum_k eps_k z_k, with 6 solves, ONE rhs evaluation and two
6-term combinations per step.  Prints Theta, beta, eps to 20 digits (computed in 60-digit arithmetic)."""
import mpmath as mp
mp.mp.dps = 60
g = mp.mpf('0.25')
a = {(2,1):'0.1544000000000000e+01',
 (3,1):'0.9466785280815826e+00',(3,2):'0.2557011698983284e+00',
 (4,1):'0.3314825187068521e+01',(4,2):'0.2896124015972201e+01',(4,3):'0.9986419139977817e+00',
 (5,1):'0.1221224509226641e+01',(5,2):'0.6019134481288629e+01',(5,3):'0.1253708332932087e+02',(5,4):'-0.6878860361058950e+00'}
c = {(2,1):'-0.5668800000000000e+01',
 (3,1):'-0.2430093356833875e+01',(3,2):'-0.2063599157091915e+00',
 (4,1):'-0.1073529058151375e+00',(4,2):'-0.9594562251023355e+01',(4,3):'-0.2047028614809616e+02',
 (5,1):'0.7496443313967647e+01',(5,2):'-0.1024680431464352e+02',(5,3):'-0.3399990352819905e+02',(5,4):'0.1170890893206160e+02',
 (6,1):'0.8083246795921522e+01',(6,2):'-0.7981132988064893e+01',(6,3):'-0.3152159432874371e+02',(6,4):'0.1631930543123136e+02',(6,5):'-0.6058818238834054e+01'}
s = 6
A = [[mp.mpf(0)]*s for _ in range(s)]; C = [[mp.mpf(0)]*s for _ in range(s)]
for (i,j),v in a.items(): A[i-1][j-1] = mp.mpf(v)
for j in range(4): A[5][j] = A[4][j]
A[5][4] = mp.mpf(1)
for (i,j),v in c.items(): C[i-1][j-1] = mp.mpf(v)
m = [A[4][0], A[4][1], A[4][2], A[4][3], mp.mpf(1), mp.mpf(1)]
Theta = []
for i in range(s):
    th = [mp.mpf(0)]*(s+1)
    th[0] = g                       # gamma * z_1
    for j in range(i):
        coef = A[i][j] + g*C[i][j]
        for k in range(s):
            th[k+1] += coef*Theta[j][k]          # M^{-1} u_j : shift
            th[k] -= A[i][j]*Theta[j][k]
    Theta.append(th[:s])
beta = [sum(m[i]*Theta[i][k] for i in range(s)) for k in range(s)]
eps = Theta[5]
if __name__ == '__main__':
    for i in range(s): print('Theta[%d] =' % (i+1), [mp.nstr(x, 8) for x in Theta[i]])
    print('beta =', [mp.nstr(x, 20) for x in beta])
    print('eps  =', [mp.nstr(x, 20) for x in eps])
    # sanity: stability function R(z) = 1 + sum_k beta_k z / (1 - gamma z)^k must match exp(z) to O(z^5)
    for z in (mp.mpf('0.1'), mp.mpf('-0.2'), mp.mpf('-1e6')):
        R = 1 + sum(beta[k]*z/(1-g*z)**(k+1) for k in range(s))
        E = sum(eps[k]*z/(1-g*z)**(k+1) for k in range(s))
        print('z =', mp.nstr(z,5), ' R(z) - exp(z) =', mp.nstr(R - mp.e**z, 5), ' err fn =', mp.nstr(E,5), ' z^5 =', mp.nstr(z**5, 5))
