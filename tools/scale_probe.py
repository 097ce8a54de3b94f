import sys, os, time, ctypes as C
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from biscuit_amd import _lib as B
from biscuit_amd.api import Index, default_opt
import oracle_lib
L=B.lib()
work='/tmp/probe'; os.makedirs(work, exist_ok=True)
B.check(L.bsx_sim_genome((work+'/g.fa').encode(), C.c_int64(8_000_000), C.c_uint64(5), 4, C.c_double(0.05)),'g')
idx=Index.build(work+'/g.fa', work+'/g')
L.bsx_sim_pairs.argtypes=[C.c_void_p,C.c_int64,C.c_int,C.c_uint64,C.c_int,C.c_int,C.c_double,C.c_double,C.POINTER(C.c_void_p)]
L.bsx_process_seqs_backend.argtypes=[C.c_void_p]*3+[C.c_int64,C.c_int,C.c_void_p,C.c_void_p]
L.bsx_sim_reset_reads.argtypes=[C.c_void_p,C.c_int64]
n=100000
p=C.c_void_p(); B.check(L.bsx_sim_pairs(idx.h,n,150,3,200,500,0.005,0.0,C.byref(p)),'p')
opt=default_opt(); opt.flag|=0x12; opt.n_threads=16
C.c_int.in_dll(L,'bsx_verbose').value=3
os.environ['BSX_PHASES']='1'
for nt in (1,4,16):
    os.environ['BSX_HOST_THREADS']=str(nt)
    port=oracle_lib.Port(idx, n_threads=nt); be=port.backend()
    t=time.time(); B.check(L.bsx_process_seqs_backend(C.byref(be),C.byref(opt),idx.h,0,2*n,p,None),'x'); print('threads',nt,'wall %.2f'%(time.time()-t), flush=True)
    L.bsx_sim_reset_reads(p,2*n)
