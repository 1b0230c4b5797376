"""ARK4(3)6L[2]SA (Kennedy & Carpenter, Appl. Numer. Math. 44 (2003)): the additive Runge-Kutta pair (explicit + L-stable ESDIRK, gamma = 1/4,\n6 stages, order 4, embedded order 3) checked against EVERY additive order condition up to order 4 -- incl. the coupling conditions that mix\nthe two tableaus -- in exact rational arithmetic (dev tool; residuals <= 3e-26: the explicit entries are 13-digit rational approximations)."""
from fractions import Fraction as Fr
import itertools
g = Fr(1,4)
c = [Fr(0), Fr(1,2), Fr(83,250), Fr(31,50), Fr(17,20), Fr(1)]
AI = [[0]*6 for _ in range(6)]
AI[1][0]=Fr(1,4); AI[1][1]=g
AI[2][0]=Fr(8611,62500); AI[2][1]=Fr(-1743,31250); AI[2][2]=g
AI[3][0]=Fr(5012029,34652500); AI[3][1]=Fr(-654441,2922500); AI[3][2]=Fr(174375,388108); AI[3][3]=g
AI[4][0]=Fr(15267082809,155376265600); AI[4][1]=Fr(-71443401,120774400); AI[4][2]=Fr(730878875,902184768); AI[4][3]=Fr(2285395,8070912); AI[4][4]=g
b=[Fr(82889,524892),Fr(0),Fr(15625,83664),Fr(69875,102672),Fr(-2260,8211),g]
AI[5]=list(b)
bh=[Fr(4586570599,29645900160),Fr(0),Fr(178811875,945068544),Fr(814220225,1159782912),Fr(-3700637,11593932),Fr(61727,225920)]
AE=[[Fr(0)]*6 for _ in range(6)]
AE[1][0]=Fr(1,2)
AE[2][0]=Fr(13861,62500); AE[2][1]=Fr(6889,62500)
AE[3][0]=Fr(-116923316275,2393684061468); AE[3][1]=Fr(-2731218467317,15368042101831); AE[3][2]=Fr(9408046702089,11113171139209)
AE[4][0]=Fr(-451086348788,2902428689909); AE[4][1]=Fr(-2682348792572,7519795681897); AE[4][2]=Fr(12662868775082,11960479115383); AE[4][3]=Fr(3355817975965,11060851509271)
AE[5][0]=Fr(647845179188,3216320057751); AE[5][1]=Fr(73281519250,8382639484533); AE[5][2]=Fr(552539513391,3454668386233); AE[5][3]=Fr(3354512671639,8306763924573); AE[5][4]=Fr(4040,17871)
AI=[[Fr(x) for x in r] for r in AI]
def mv(A,v): return [sum(A[i][j]*v[j] for j in range(6)) for i in range(6)]
def dot(u,v): return sum(x*y for x,y in zip(u,v))
one=[Fr(1)]*6
print("row sums I:", [float(sum(AI[i])-c[i]) for i in range(6)])
print("row sums E:", [float(sum(AE[i])-c[i]) for i in range(6)])
for name,w in (("b",b),("bhat",bh)):
    print(name, "sum-1", float(dot(w,one)-1), "b.c-1/2", float(dot(w,c)-Fr(1,2)), "b.c2-1/3", float(dot(w,[x*x for x in c])-Fr(1,3)))
    for s,A in (("I",AI),("E",AE)):
        print("  ",name,"b.A%s.c-1/6"%s, float(dot(w,mv(A,c))-Fr(1,6)))
    print("  b.c3-1/4", float(dot(w,[x**3 for x in c])-Fr(1,4)))
    for s,A in (("I",AI),("E",AE)):
        Ac=mv(A,c)
        print("  b.(c*A%sc)-1/8"%s, float(dot(w,[ci*x for ci,x in zip(c,Ac)])-Fr(1,8)), " b.A%s.c2-1/12"%s, float(dot(w,mv(A,[x*x for x in c]))-Fr(1,12)))
    for (s1,A1),(s2,A2) in itertools.product((("I",AI),("E",AE)),repeat=2):
        print("  b.A%s.A%s.c-1/24"%(s1,s2), float(dot(w,mv(A1,mv(A2,c)))-Fr(1,24)))
