import sys, time, torch, numpy as np
sys.path.insert(0, '.')
import pytorch_sparse_amd as ts
dev = torch.device('cuda:0')
z = np.load('tests/golden/py7_c1_spmm.npz')
m, n = int(z['m']), int(z['n'])
index = torch.from_numpy(z['index']).to(dev); value = torch.from_numpy(z['value']).to(dev); x = torch.from_numpy(z['mat']).to(dev)
for _ in range(3): out = ts.spmm(index, value, m, n, x)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): ts.spmm(index, value, m, n, x)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out_g = ts.spmm(index, value, m, n, x)
g.replay(); torch.cuda.synchronize()
print('equal', torch.equal(out_g, out), float((out_g - torch.from_numpy(z['out']).to(dev)).abs().max()))
def wall(fn, it=200):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
print('eager pipelined ms', wall(lambda: ts.spmm(index, value, m, n, x)))
print('graph replay ms', wall(g.replay))
# new values, same graph
value.mul_(2.0); g.replay(); torch.cuda.synchronize()
print('after update equal 2x', torch.allclose(out_g, 2 * out, rtol=1e-6))
