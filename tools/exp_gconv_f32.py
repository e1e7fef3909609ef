import torch, torch.nn.functional as F, time
d=torch.device('cuda:0')
B,H,W,G,hc,km=4,180,180,42,64,3
y=torch.randn(B,G*hc,H,W,device=d).contiguous(memory_format=torch.channels_last).requires_grad_(True)
w=(torch.randn(G*km,hc,3,3,device=d)*0.05).requires_grad_(True)
wd=torch.zeros(128,G*hc,3,3,device=d).contiguous(memory_format=torch.channels_last).requires_grad_(True)
def t(fn,n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
def grouped():
    z=F.conv2d(y,w,None,padding=1,groups=G); z.sum().backward()
def dense():
    z=F.conv2d(y,wd,None,padding=1); z.sum().backward()
def grouped_nchw():
    yy=y.detach().contiguous().requires_grad_(True)
    z=F.conv2d(yy,w,None,padding=1,groups=G); z.sum().backward()
print('grouped fwd+bwd ms', t(grouped)); print('dense blockdiag fwd+bwd ms', t(dense))
with torch.no_grad():
    print('grouped fwd ms', t(lambda: F.conv2d(y,w,None,padding=1,groups=G))); print('dense fwd ms', t(lambda: F.conv2d(y,wd,None,padding=1)))
