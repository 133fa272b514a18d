import sys, time, torch
sys.path.insert(0, '/root/repo')
import agilex_ntt_amd as agx
n=4096; q=agx.find_primes(60,n)[0]; plan=agx.Plan(n,[q]); st=torch.cuda.current_stream().cuda_stream
d=torch.empty(n,dtype=torch.int64,device='cuda'); plan.fill_synthetic(d.data_ptr(),1,0,42,st)
for _ in range(20): plan.forward(d.data_ptr(),d.data_ptr(),1,st); plan.inverse(d.data_ptr(),d.data_ptr(),1,st)
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): plan.forward(d.data_ptr(),d.data_ptr(),1,st); plan.inverse(d.data_ptr(),d.data_ptr(),1,st)
e1.record(); torch.cuda.synchronize()
print("config 2: n=4096 one 60-bit modulus batch=1: forward+inverse pair %.2f us (device time per pair, back-to-back launches)" % (e0.elapsed_time(e1)/200*1000))
t0=time.perf_counter()
for _ in range(200):
    plan.forward(d.data_ptr(),d.data_ptr(),1,st); plan.inverse(d.data_ptr(),d.data_ptr(),1,st); torch.cuda.synchronize()
print("with a host sync per pair: %.2f us" % ((time.perf_counter()-t0)/200*1e6))
g=torch.cuda.CUDAGraph(); side=torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        s=torch.cuda.current_stream().cuda_stream
        for _ in range(10): plan.forward(d.data_ptr(),d.data_ptr(),1,s); plan.inverse(d.data_ptr(),d.data_ptr(),1,s)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
e0.record()
for _ in range(20): g.replay()
e1.record(); torch.cuda.synchronize()
print("hipGraph of 10 pairs: %.2f us per pair" % (e0.elapsed_time(e1)/200*1000))
