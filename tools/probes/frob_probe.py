import torch, time, sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nn_fac_amd.engine import get_engine
eng = get_engine("cuda:0")
I = J = 500; R = 30
for K in (64, 128, 256, 500, 1000, 2000):
    T = torch.rand(I, J, K, device="cuda")
    F = [torch.rand(R, d, device="cuda") for d in (I, J, K)]
    cost = torch.zeros(1, dtype=torch.float64, device="cuda"); Y = torch.empty(R, I, J, device="cuda")
    for _ in range(3): eng.cp3_partial_cost(T, F, Y, cost)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): eng.cp3_partial_cost(T, F, Y, cost)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 4.0 * R * I * J * K
    print(f"K={K:5d}: {us:8.1f} us/call  {fl / us / 1e6:6.1f} TF  {4.0*I*J*K/us/1e6:6.2f} TB/s")
