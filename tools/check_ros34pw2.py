"""Verify the ROS34PW2 table (Rang & Angermann, BIT 45 (2005): 4-stage, order 3, stiffly accurate ROSENBROCK-W method, i.e. the
order conditions hold for an ARBITRARY approximation of the Jacobian) against the order-3 W-method conditions in 50-digit
arithmetic (dev tool).  Form: (A, Gamma, b, bhat) with Gamma including the diagonal gamma."""
import mpmath as mp
mp.mp.dps = 50
M = mp.mpf
g = M('4.3586652150845900e-01')
A = mp.matrix([[0,0,0,0],[M('8.7173304301691801e-01'),0,0,0],[M('8.4457060015369423e-01'),M('-1.1299064236484185e-01'),0,0],[0,0,1,0]])
G = mp.matrix([[g,0,0,0],[M('-8.7173304301691801e-01'),g,0,0],[M('-9.0338057013044082e-01'),M('5.4180672388095326e-02'),g,0],
               [M('2.4212380706095346e-01'),M('-1.2232505839045147e+00'),M('5.4526025533510214e-01'),g]])
b = mp.matrix([M('2.4212380706095346e-01'),M('-1.2232505839045147e+00'),M('1.5452602553351020e+00'),g]).T
bh = mp.matrix([M('3.7810903145819369e-01'),M('-9.6042292212423178e-02'),M('5.0000000000000000e-01'),M('2.1793326075422950e-01')]).T
one = mp.matrix([1,1,1,1])
def conds(b):
    a1 = A*one; g1 = G*one
    return {
        'b.1 = 1': (b*one)[0] - 1,
        'b.A1 = 1/2': (b*a1)[0] - M(1)/2,
        'b.G1 = 0': (b*g1)[0],
        'b.(A1)^2 = 1/3': sum(b[i]*a1[i]**2 for i in range(4)) - M(1)/3,
        'b.A.A1 = 1/6': (b*A*a1)[0] - M(1)/6,
        'b.A.G1 = 0': (b*A*g1)[0],
        'b.G.A1 = 0': (b*G*a1)[0],
        'b.G.G1 = 0': (b*G*g1)[0],
    }
print('main weights (order 3 W-method: all eight must vanish)')
for k, v in conds(b).items(): print('  %-16s %s' % (k, mp.nstr(v, 5)))
print('embedded weights (order 2 W: first three must vanish)')
for k, v in conds(bh).items(): print('  %-16s %s' % (k, mp.nstr(v, 5)))
# stiff accuracy: last row of A + Gamma equals b
print('stiffly accurate: max |(A+G)[3,:] - b| =', mp.nstr(max(abs((A+G)[3,j]-b[j]) for j in range(4)), 5))
# stability function with exact Jacobian: R(z) = 1 + z b (I - z(A+G))^-1 1 ; R(inf)
B = A + G
print('R(inf) =', mp.nstr(1 - (b*(B**-1)*one)[0], 5), '  Rhat(inf) =', mp.nstr(1 - (bh*(B**-1)*one)[0], 5))

# ---- implementation form:  (I/(h gamma) - J~) U_i = f(y + sum_j a_ij U_j) + sum_j (c_ij / h) U_j ,  y1 = y + sum m_i U_i
Ginv = G**-1
a = A*Ginv
c = mp.diag([1/g]*4) - Ginv
m = b*Ginv; mh = bh*Ginv
print('\nimplementation form (constexpr tables for csrc/pk_network_solve.hpp):')
print('GAM =', mp.nstr(g, 20))
for i in range(1, 4):
    print('a%d = {' % (i+1) + ', '.join(mp.nstr(a[i, j], 20) for j in range(i)) + '}')
for i in range(1, 4):
    print('c%d = {' % (i+1) + ', '.join(mp.nstr(c[i, j], 20) for j in range(i)) + '}')
print('m  = {' + ', '.join(mp.nstr(m[j], 20) for j in range(4)) + '}')
print('e  = {' + ', '.join(mp.nstr(m[j] - mh[j], 20) for j in range(4)) + '}   (m - mhat)')
