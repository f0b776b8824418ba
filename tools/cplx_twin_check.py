import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from russell_amd import problems as P
from russell_amd.backend import Hipmf
import scipy.sparse as sp
lib=sys.argv[1] if len(sys.argv)>1 else None
n, rp, ci, v = P.brusselator_pattern(9)
A = sp.csr_matrix((v,ci,rp),shape=(n,n)).tocoo()
al, be = 2.68e4, 3.05e4
rows=[];cols=[];vals=[]
for i,j,a in zip(A.row,A.col,A.data):
    a = -a + (al if i==j else 0.0); b = be if i==j else 0.0
    rows += [2*i,2*i,2*i+1,2*i+1]; cols += [2*j,2*j+1,2*j,2*j+1]; vals += [a,-b,b,a]
K = sp.csr_matrix((vals,(rows,cols)),shape=(2*n,2*n)); K.sum_duplicates(); K.sort_indices()
N=2*n
xs = np.random.default_rng(1).standard_normal(N)
b = K@xs
for fused in ("1","0"):
    os.environ["HIPMF_FUSED_SOLVE"]=fused
    s=Hipmf(lib); assert s.initialize(N,K.indptr.astype(np.int32),K.indices.astype(np.int32),refinement_nstep=0,values=K.data)==0
    assert s.factorize(K.data)==0
    x=s.solve(b); st=s.stats()
    print("fused",fused,"err",np.max(np.abs(x-xs)),"maxfront",st["max_front"],"nsuper",st["nsuper"],"levels",st["nlevels"])
    s.close()
