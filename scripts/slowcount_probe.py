import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from apples_amd import synth
from apples_amd.engine import Engine
d=synth.make_dataset(10000,1000,10000)
nodes=np.array([d.tree.name_to_node[n] for n in d.ref_names],np.int32)
eng=Engine(d.tree,d.ref_seqs,nodes,method='OLS')
cnt,dist=eng.distances(d.query_seqs[:2000])
within=((dist>=0)&(dist<=0.2)).sum(1)
print('queries',len(within),'with <25 within threshold:',(within<25).sum(),'min',within.min(),'median',np.median(within),'max',within.max())
