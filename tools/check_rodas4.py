"""Verify the RODAS4 (Hairer & Wanner, 'Solving ODEs II', rodas.f METH=1) coefficient
table used by the HIP integrator against the Rosenbrock order conditions, in 50-digit
arithmetic.  Dev tool; prints residuals (should be ~1e-16, limited by the 16-digit table)."""
import mpmath as mp
mp.mp.dps = 50
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
A = mp.zeros(s, s); C = mp.zeros(s, s)
for (i,j),v in a.items(): A[i-1,j-1] = mp.mpf(v)
for j in range(4): A[5,j] = A[4,j]
A[5,4] = 1
for (i,j),v in c.items(): C[i-1,j-1] = mp.mpf(v)
m = mp.matrix([A[4,0],A[4,1],A[4,2],A[4,3],1,1]).T
mh = mp.matrix([A[4,0],A[4,1],A[4,2],A[4,3],1,0]).T
Ginv = mp.diag([1/g]*s) - C
G = Ginv**-1
alpha = A*G
b = m*G; bh = mh*G
gam = G - mp.diag([g]*s)          # strictly lower gamma_ij
beta = alpha + gam
def S(f): return mp.fsum(f)
ai = [S(alpha[i,j] for j in range(s)) for i in range(s)]
bi = [S(beta[i,j] for j in range(s)) for i in range(s)]
def conds(b):
    r = {}
    r['1'] = S(b[i] for i in range(s)) - 1
    r['2'] = S(b[i]*bi[i] for i in range(s)) - (mp.mpf(1)/2 - g)
    r['3a'] = S(b[i]*ai[i]**2 for i in range(s)) - mp.mpf(1)/3
    r['3b'] = S(b[i]*beta[i,j]*bi[j] for i in range(s) for j in range(s)) - (mp.mpf(1)/6 - g + g*g)
    r['4a'] = S(b[i]*ai[i]**3 for i in range(s)) - mp.mpf(1)/4
    r['4b'] = S(b[i]*ai[i]*alpha[i,k]*bi[k] for i in range(s) for k in range(s)) - (mp.mpf(1)/8 - g/3)
    r['4c'] = S(b[i]*beta[i,j]*ai[j]**2 for i in range(s) for j in range(s)) - (mp.mpf(1)/12 - g/3)
    r['4d'] = S(b[i]*beta[i,j]*beta[j,k]*bi[k] for i in range(s) for j in range(s) for k in range(s)) - (mp.mpf(1)/24 - g/2 + 3*g*g/2 - g**3)
    return r
print('main weights b (order 4 expected):')
for k,v in conds(b).items(): print('  cond', k, mp.nstr(v, 5))
print('embedded weights bhat (order 3 expected; 4x residuals nonzero):')
for k,v in conds(bh).items(): print('  cond', k, mp.nstr(v, 5))
print('c_i =', [mp.nstr(x, 8) for x in ai])
print('gamma_i sum =', [mp.nstr(S(G[i,j] for j in range(s)),8) for i in range(s)])
# stability function at infinity: R(inf) = 1 - b^T Gamma^-1 1  (should be 0: stiffly accurate)
one = mp.matrix([1]*s)
print('R(inf) =', mp.nstr(1 - (b*Ginv*one)[0], 5), ' Rhat(inf)=', mp.nstr(1 - (bh*Ginv*one)[0], 5))
