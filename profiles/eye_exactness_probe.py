import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, cases
for name in ("eye_256","multi_draw_320x200"):
    c=cases.CASES[name]()
    fb,z,st,line=cases.run_gpu(c); ofb,oz,ost=cases.run_oracle(c)
    d=np.abs(fb.astype(int)-ofb.astype(int))
    print(name,'max diff',d.max(),'differing bytes',(d>0).sum(),'of',d.size)
