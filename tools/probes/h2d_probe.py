import torch, time
dev=torch.device("cuda:0")
n=36*1024*1024
h=torch.empty(n,dtype=torch.uint8,pin_memory=True); d=torch.empty(n,dtype=torch.uint8,device=dev)
s=torch.cuda.Stream()
def t(reps=10):
    torch.cuda.synchronize(); t0=time.perf_counter()
    with torch.cuda.stream(s):
        for _ in range(reps): d.copy_(h,non_blocking=True)
    s.synchronize(); return (time.perf_counter()-t0)/reps*1e3
t(); print("idle H2D 36MB ms:", round(t(),3))
a=torch.randn(8192,8192,device=dev)
def busy():
    for _ in range(20): torch.mm(a,a)
busy(); torch.cuda.synchronize()
t0=time.perf_counter(); busy(); x=t(); torch.cuda.synchronize(); print("H2D beside GEMMs ms:", round(x,3))
