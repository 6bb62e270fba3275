import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.kbench import timeit
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ln_lab.so"))
lib.run.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p]
rows = 21920
x = torch.randn(rows, 1024, device="cuda"); w = torch.ones(1024, device="cuda"); b = torch.zeros(1024, device="cuda")
out = torch.empty(rows, 1024, device="cuda", dtype=torch.bfloat16); dummy = torch.zeros(4, device="cuda")
names = {0: "full", 1: "no reductions", 2: "no store", 3: "loads only", 4: "16-B stores (lane owns 8 ch)"}
for v in range(5):
    med, _ = timeit(lambda: lib.run(v, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), dummy.data_ptr(), rows, torch.cuda.current_stream().cuda_stream), iters=20)
    print(v, names[v], round(med * 1e3, 1), "us")
