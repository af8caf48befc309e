#!/usr/bin/env python3
"""Paired-end 2x150 reads against the GTDB-scale synthetic index (16-plane counters, one-wave sort+unique): K1/K2 kernel times."""
import sys, torch, numpy as np
sys.path.insert(0,'.')
from kmcp_amd import Database, default_params, lib
dev=torch.device('cuda:0')
spec=lib.SynthSpec(k=21,num_hashes=1,fpr=0.3,n_blocks=32,cols_per_block=14976,num_sigs=968700,kmers_per_col=345510,seed=42)
db=Database.open_synthetic(spec,device=0); db.set_profiling(True)
B=131072; L=150
g=torch.Generator(device=dev); g.manual_seed(1)
acgt=torch.tensor(list(b"ACGT"),dtype=torch.uint8,device=dev)
r1=acgt[torch.randint(0,4,(B,L),generator=g,device=dev)].contiguous().view(-1)
r2=acgt[torch.randint(0,4,(B,L),generator=g,device=dev)].contiguous().view(-1)
offs=(torch.arange(B+1,device=dev,dtype=torch.int64)*L).contiguous()
cap=8*B
hits=torch.zeros((cap,3),dtype=torch.int32,device=dev); cnt=torch.zeros(2,dtype=torch.int64,device=dev)
qk=torch.zeros(B,dtype=torch.int32,device=dev); ql=torch.zeros(B,dtype=torch.int32,device=dev)
for mode in ("SE","PE"):
    for i in range(3):
        db.query_device(r1.data_ptr(),offs.data_ptr(),B,B*L*(2 if mode=="PE" else 1),L,hits.data_ptr(),cap,cnt.data_ptr(),qk.data_ptr(),ql.data_ptr(),params=default_params(),
                        d_seqs2=r2.data_ptr() if mode=="PE" else None, d_offs2=offs.data_ptr() if mode=="PE" else None)
        torch.cuda.synchronize()
        a,b=db.last_timing()
    nk=int(qk.sum().item())
    byts=nk*32*1872
    print(mode,"kmers_ms %.2f cobs_ms %.1f  kmers/query %.0f  algorithmic %.2f TB/s"%(a,b,nk/B,byts/b/1e9))
