#!/usr/bin/env python3
"""Time-bounded random soak of the whole HOST half of GPSIQ_NCO_REFERENCE on the CPU: gpsiq_reference_batch (carrier chain, both
wrap-to-wrap walks, candidate search, patches) + the oracle's closed form + the patches applied == the float loop
(oracle_block_float), every element and the carried phase; scenarios with phases a hair off LUT / chip boundaries and whole
numbers of samples per chip, so that there are many candidates and patches.  usage: python tests/soak_reference_host.py seed seconds"""
import sys, time
import os; HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'multi-sdr-gps-sim_amd'))
import numpy as np, gpsiq, _oracle
from _oracle import apply_patches
from gpsiq.abi import SC08, SC16
from gpsiq.scenario import synth_blocks
o=_oracle.load_oracle()
rng=np.random.default_rng(int(sys.argv[1]))
t0=time.time(); runs=0; npatch=0
while time.time()-t0 < float(sys.argv[2]):
    fs=float(rng.choice([2.6e6,3e6,10e6,25e6,1.2e6,2.048e6]))
    ns=int(rng.integers(1000, int(fs)//10+1)) if rng.random()<0.5 else int(fs)//10
    ns=min(ns, 600000)
    nb=int(rng.integers(1,4)); nc=int(rng.integers(1,17))
    d=synth_blocks(nb,nc,seed=int(rng.integers(1<<30)))
    if rng.random()<0.5:
        k=rng.integers(0,512,nc)
        d["carr_phase"][0]=(k+rng.choice([1e-13,-1e-13,3e-12,0.0],nc))/512.0%1.0
        d["code_phase"][:]=(rng.integers(0,1023,(nb,nc))+rng.choice([1e-10,2e-9,0.0,1.0-1e-10],(nb,nc)))%1023.0
    if rng.random()<0.3:
        d["f_code"][:, :max(1,nc//2)]=fs/rng.integers(3,30,max(1,nc//2))
    ss=int(rng.choice([SC08,SC16]))
    want=[];carr=None;prev=None
    for b in range(nb):
        db=d[b].copy()
        if b: db["carr_phase"]=np.where((prev==db["prn"])&(db["prn"]>0),carr,db["carr_phase"])
        w,carr=o.block_float(db,ns,fs,ss); want.append(w); prev=db["prn"].copy()
    q,patches,cend=gpsiq.reference_blocks(d,fs,ns)
    for b in range(nb):
        g=o.block_fixed(q[b],ns,ss,seq=True)
        apply_patches(o,q[b],g,patches[patches["block"]==b],ss)
        if not np.array_equal(g,want[b]):
            print("MISMATCH",fs,ns,nb,nc,ss,b); sys.exit(1)
    act=d[-1]["prn"]>0
    if not np.array_equal(cend[act],carr[act]): print("CARR MISMATCH"); sys.exit(1)
    runs+=1; npatch+=len(patches)
st=gpsiq.reference_stats()
print(runs,"runs",npatch,"patches, all equal to the float loop; candidate states %d, decided from the start state %d, walked: carrier %d code %d"%st)
