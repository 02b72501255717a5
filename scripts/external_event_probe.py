import torch, time, inspect
print(torch.__version__)
print('external' in inspect.signature(torch.cuda.Event.__new__).parameters if hasattr(torch.cuda.Event,'__new__') else None, torch.cuda.Event.__doc__[:600] if torch.cuda.Event.__doc__ else None)
try:
    e_ready = torch.cuda.Event(external=True)
    e_done = torch.cuda.Event(external=True)
except TypeError as ex:
    print('no external events:', ex); raise SystemExit
x = torch.zeros(1<<20, device='cuda'); y = torch.zeros_like(x)
side = torch.cuda.Stream()
cap = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(cap):
    x.add_(1)      # warm
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=cap, capture_error_mode='thread_local'):
    x.add_(1.0)                 # producer
    e_ready.record()            # external record node
    y.add_(1.0)                 # independent work
    e_done.wait()               # external wait node
    x.mul_(2.0)                 # consumer: must see the side stream's update
torch.cuda.synchronize()
x.zero_(); y.zero_()
torch.cuda.synchronize()
for it in range(3):
    g.replay()
    with torch.cuda.stream(side):
        e_ready.wait()          # eager wait on the event recorded inside the graph
        x.add_(10.0)            # "all-reduce"
        e_done.record()
    torch.cuda.synchronize()
    print(it, float(x[0]), float(y[0]))
# expected: it0: (0+1+10)*2 = 22; it1: (22+1+10)*2=66; it2: (66+11)*2=154
