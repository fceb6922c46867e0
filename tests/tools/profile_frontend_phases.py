import sys, numpy as np
import os; ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from mercury_amd import RxPhy
import oraclelib
for cfg in (8,0,16):
    o=oraclelib.Oracle(cfg); F=2048
    rx=RxPhy(cfg,max_batch=F, agc=1 if cfg!=16 else 0, variance_source=1 if cfg!=16 else 0)
    bb=np.stack([o.gen_frame(1,i,oraclelib.noise_amp_for(5.0))[0] for i in range(8)])
    bb=np.tile(bb,(F//8,1))
    out=rx.receive(bb,taps=True); out=rx.receive(bb,taps=True)
    c=out['cycles']; d=np.diff(c[:9])
    names=['fft','agc','estimate','mean_H tap','pilot cells','variance sums | data cells','demap','repack']
    print(cfg,'total cycles',c[8]-c[0], dict(zip(names,d.tolist())))
