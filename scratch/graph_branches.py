import torch, time
def run(nstreams, chain=50, n=4096):
    xs=[torch.zeros(n, device="cuda") for _ in range(4)]
    streams=[torch.cuda.Stream() for _ in range(nstreams)]
    def work():
        cur=torch.cuda.current_stream()
        if nstreams==1:
            for x in xs:
                for _ in range(chain): x.add_(1)
            return
        for i,x in enumerate(xs):
            s=streams[i%nstreams]; s.wait_stream(cur)
            with torch.cuda.stream(s):
                for _ in range(chain): x.add_(1)
        for s in streams: cur.wait_stream(s)
    side=torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): work()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): work()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
    print(f"streams={nstreams}: {dt*1e6:.0f} us per replay of {4*chain} tiny kernels ({dt*1e6/(4*chain):.2f} us/kernel), check {xs[0][0].item()}")
for ns in (1,2,4): run(ns)
