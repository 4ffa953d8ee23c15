import torch, time
dev=torch.device("cuda:0")
torch.cuda.set_device(0)
main=torch.cuda.current_stream(dev)
def overlaps(a,b,cycles=3_000_000):
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    f0=torch.cuda.Event(enable_timing=True); f1=torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        e0.record(); torch.cuda._sleep(cycles); e1.record()
    with torch.cuda.stream(b):
        f0.record(); torch.cuda._sleep(cycles); f1.record()
    torch.cuda.synchronize()
    single=e0.elapsed_time(e1)
    span=max(e0.elapsed_time(e1), e0.elapsed_time(f1))
    return single, span
c=[torch.cuda.Stream(device=dev) for _ in range(10)]
overlaps(main,c[0])
for i,s in enumerate(c):
    overlaps(main,s)
    print(i, overlaps(main,s), overlaps(main,s))
hp=torch.cuda.Stream(device=dev, priority=-1)
overlaps(main,hp)
print('prio', overlaps(main,hp), overlaps(main,hp))
print('c0 vs c1', overlaps(c[0],c[1]), 'c0 vs c3', overlaps(c[0],c[3]))
