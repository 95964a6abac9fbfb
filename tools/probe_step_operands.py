import sys, time, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uvs_amd import engine, _lib
import bench
des = bench.config2()['experiments']['desired_f']
rng=np.random.default_rng(1)
T=1
fp = engine.make_params(8, 6, 'GMCKF', 10, False, 0.05, 15, 0.2, des, True, 0, 0)
fn=_lib.lib().uvs_rmckf_step_f64
def mk(shape, host, dtype=torch.float64):
    t=torch.zeros(shape,dtype=dtype)
    return t.pin_memory() if host else t.cuda()
for in_host in (False, True):
    for out_host in (False, True):
        bank = engine.FilterBank(fp, T, rng.normal(size=(T, 48)) * 50, 'cuda')
        f=mk((T,8),in_host); f_old=mk((T,8),in_host); dqp=mk((T,6),in_host)
        f.copy_(torch.as_tensor(np.asarray(des)[None]+rng.normal(size=(T,8)))); f_old.copy_(f+0.1)
        dq=mk((T,6),out_host); err=mk((T,8),out_host); kap=mk((T,8),out_host); st=mk((T,),out_host,torch.int32)
        s=C.c_void_p(torch.cuda.current_stream().cuda_stream)
        call=lambda k: fn(C.byref(fp),T,bank.X.data_ptr(),bank.P.data_ptr(),f.data_ptr(),f_old.data_ptr(),dqp.data_ptr(),0,k,dq.data_ptr(),err.data_ptr(),kap.data_ptr(),st.data_ptr(),s)
        for k in range(20): call(k)
        torch.cuda.synchronize()
        t0=time.perf_counter()
        for k in range(200): call(k%299)
        torch.cuda.synchronize()
        print(f'inputs {"host" if in_host else "dev "} outputs {"host" if out_host else "dev "}: {(time.perf_counter()-t0)/200*1e6:.1f} us per step back to back')
